"""Empty stand-in for third-party cv2 (absent here)."""
