from .trajectories import get_optical_flow_tile_mask, coeffs_grid_to_list  # noqa: F401
from .basis import compute_basis, basis_values, bernstein_basis, trajectories_from_bezier, bspline_basis, trajectories_from_bspline  # noqa: F401
from .event_image_converter import EventImageConverter  # noqa: F401
from .voxel_grid import VoxelGrid, voxel_grids  # noqa: F401
from .ingest import ingest_events  # noqa: F401
from .flow import dense_flow_from_traj, calculate_flow_error, ErrorCalculatorFactory, OpticalFlowError  # noqa: F401
