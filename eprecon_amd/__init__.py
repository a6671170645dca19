"""eprecon_amd — MI355X-native per-fragment 3D path of EPRecon behind the reference's module API.

Modules mirror the reference files they replace:
  back_project.py             ops/back_project.py + Back_Project (models/occupancy_initialization.py)
  generate_grids.py           ops/generate_grids.py
The compute lives in csrc/*.hip behind the C ABI of include/eprecon_hip.h (libeprecon_hip.so).
"""
__version__ = "0.1.0"
