"""Mirror of utils/calibration_tools (rectification between voxelizer and model)."""
