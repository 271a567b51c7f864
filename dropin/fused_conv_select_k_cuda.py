"""Drop-in for the reference's compiled extension `fused_conv_select_k_cuda`
(src/projectPN/fused_conv_select/fused_conv_g.cpp:69-72).  Put this file's directory on
sys.path in place of the CUDA build; `src/projectPN/fused_conv_select/fused_conv_select_k.py`
then imports it unchanged and runs on libi2p_ops.so (MI355X)."""
from i2pnet_amd import ops as _ops


def fused_conv_select_k(xyz_tensor, xyz2_tensor, idx_n2_tensor, random_hw_tensor, H, W, npoints, kernel_size_H,
                        kernel_size_W, K, flag, distance, stride_h, stride_w, select_b_idx_tensor,
                        select_h_idx_tensor, select_w_idx_tensor, valid_idx_tensor, valid_in_dis_idx_tensor,
                        select_mask_tensor, small_h, small_w):
    _ops.hip_backend().fused_conv_select_k(
        xyz_tensor, xyz2_tensor, idx_n2_tensor, random_hw_tensor, H, W, npoints, kernel_size_H, kernel_size_W, K, flag,
        distance, stride_h, stride_w, select_b_idx_tensor, select_h_idx_tensor, select_w_idx_tensor, valid_idx_tensor,
        valid_in_dis_idx_tensor, select_mask_tensor, small_h, small_w)
