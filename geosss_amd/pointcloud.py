"""The reference keeps its point clouds in geosss/pointcloud.py; here they live beside the registration targets
(registration.py).  This module keeps the reference's import path working:

    from geosss_amd.pointcloud import PointCloud, RotationMatrix, RotationProjection, matrix2quat, quat2matrix
"""
from .registration import PointCloud, RotationMatrix, RotationProjection, matrix2quat, quat2matrix

__all__ = ["PointCloud", "RotationMatrix", "RotationProjection", "matrix2quat", "quat2matrix"]
