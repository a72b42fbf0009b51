"""Empty stand-in for third-party h5py (absent here; not used by the functions the goldens call)."""


class File:       # only referenced in a type annotation (loader.py:209)
    pass
