"""i2pnet_amd — MI355X (gfx950) implementation of I2PNet's point-cloud / cost-volume hot path.

Operators live in libi2p_ops.so (hand-written HIP, C ABI in include/i2p_ops.h); this package
mirrors the reference's Python operator interface on top of it:

    i2pnet_amd.pointnet2_utils          <->  pointnet2/pointnet2_utils.py
    i2pnet_amd.fused_conv_select_k      <->  src/projectPN/fused_conv_select/fused_conv_select_k.py
    i2pnet_amd.projectpn                <->  src/projectPN/utils.py
"""
__version__ = "0.1.0"
