"""Drop-in for the reference's top-level `pointnet_util.py` (imported by `src/modellearn.py:6`):
the same public names, backed by libi2p_ops.so (see i2pnet_amd/pointnet_util.py)."""
from i2pnet_amd.pointnet_util import (PointNetSetAbstraction, index_points, knn_point, sample_and_group,  # noqa: F401
                                      sample_and_group_all, square_distance)
from i2pnet_amd.pointnet2_utils import FurthestPointSampling  # noqa: F401
