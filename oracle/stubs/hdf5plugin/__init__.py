"""Empty stand-in for third-party hdf5plugin (absent here; not used by the functions the goldens call)."""
