"""Empty stand-in for third-party pandas (absent here; not used by the functions the goldens call)."""
