"""Stand-in for the third-party package pykeops==2.2.2 (absent from this image and from
/root/reference).  Only used by oracle/gen_golden.py to import the UNMODIFIED reference."""
