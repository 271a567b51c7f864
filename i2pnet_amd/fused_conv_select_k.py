"""Python operator for the projection-aware neighbour selection.

Mirror of the reference's `src/projectPN/fused_conv_select/fused_conv_select_k.py:5-26`: same
name, same 22 positional arguments, same flags, returns the six caller-allocated tensors.
"""
from . import ops

FLAG_COPY = 0b0001    # broadcast the nearest hit to all K slots (fused_conv_go.cu:211-222)
FLAG_SHIFT = 0b0010   # circular wrap along W (fused_conv_go.cu:96-113)
FLAG_FILL = 0b0100    # extension: untouched slots are written as 0, outputs need no zero-fill (include/i2p_ops.h)


def fused_conv_select_k(xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H,
                        kernel_size_W, K, flag_copy, distance, stride_h, stride_w, select_b_idx,
                        select_h_idx, select_w_idx, valid_idx, valid_in_dis_idx, select_mask,
                        small_h, small_w):
    """
    xyz1 [B,H,W,3] f32 query image, xyz2 [B,small_h,small_w,3] f32 searched image,
    idx_n2 [B,npoints,2] i32 query cells, random_hw [kH*kW] i32 window visiting order.
    Outputs (caller-allocated, caller-zeroed unless FLAG_FILL; only valid slots are written):
    select_{b,h,w}_idx [B,npoints,K,1] i64, select_mask [B,npoints,K,1] f32;
    valid_idx / valid_in_dis_idx [B,npoints,kH*kW,1] f32 are returned untouched.
    """
    ops.get_backend().fused_conv_select_k(
        xyz1, xyz2, idx_n2, random_hw, H, W, npoints, kernel_size_H, kernel_size_W, K,
        flag_copy, distance, stride_h, stride_w, select_b_idx, select_h_idx, select_w_idx,
        valid_idx, valid_in_dis_idx, select_mask, small_h, small_w)
    return select_b_idx, select_h_idx, select_w_idx, valid_idx, valid_in_dis_idx, select_mask
