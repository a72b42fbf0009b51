"""Stand-in for third-party torchvision (absent here); only gaussian_blur is functional."""
from . import transforms
