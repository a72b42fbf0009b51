"""Empty stand-in for third-party imageio (absent here; not used by the functions the goldens call)."""
