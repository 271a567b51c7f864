"""Hyper-parameters of the large-range (projection) registration model.

Same attribute names and values as the reference's config classes
(`src/config_proj_lidarcenter.py:6-150` for KITTI, `src/config_proj_lidarcenter_nus.py` for
nuScenes: dataset_type=1, init_H=21, first H stride 2) so the kernel shapes
 come out identical.  Debug/timing state of the reference classes is not carried.
"""


class I2PNetConfig:
    # every point-branch 1x1 conv is followed by a BatchNorm that ALWAYS uses batch statistics
    use_bn_p = True
    use_bn_input = True
    use_trans = True            # neighbour search on the transformed (camera-frame) cloud

    dataset_type = 0            # 0: KITTI
    rgb_encoder_channels = [    # (in, [conv channels], [max-pool strides])
        (3, [16, 16, 16, 16, 32], [2, 1, 1, 1, 2]),
        (32, [32, 32, 32, 32, 64], [2, 1, 1, 1, 2]),
        (64, [64, 64, 64, 64, 128], [1, 1, 1, 1, 2]),
    ]
    stride_Hs = [4, 2, 2, 1]
    stride_Ws = [8, 2, 2, 2]
    rank = False
    debug = False
    debug_time = False

    down_conv_dis = [0.75, 3.0, 6.0, 12.0]
    init_H = 64
    init_W = 1800
    fup = 2.0
    fdown = -24.8
    kernel_sizes = [[9, 15], [9, 15], [5, 9], [5, 9]]

    lidar_feature_size = 7
    using_intens = False
    raw_feat_point = True
    lidar_group_samples = [32, 16, 16, 16, 16]
    lidar_encoder_mlps = [
        [16, 16, 32],
        [32, 32, 64],
        [64, 64, 128],
        [128, 128, 256],
        [128, 64, 64],          # set conv that resamples the cost volume to level 4
    ]

    cost_volume_dis = [4.5, 4.5]
    cost_volume_kernel_size = [[3, 5], [3, 5]]
    cost_volume_mlps = [[128, 64, 64], [128, 64]]
    cost_volume_nsamples = [4, [-1, 32]]     # pc-stage K; pi-stage: all pixels / 32-NN pixels
    backward_validation = [True, False]

    up_conv_dis = [9.0, 9.0]
    up_conv_kernel_size = [[5, 9], [5, 9]]
    setupconv_mlps = [[[128, 64], [64]], [[128, 64], [64]]]
    setupconv_nsamples = [8, 8]
    flow_predictor_mlps = [[128, 64], [128, 64], [128, 64]]

    head_hidden_dim = 256
    rotation_quat_head_dim = 4
    transition_vec_head_dim = 3
    head_dropout_rate = 0.5
    head_pos_embedding = False
    split_dp = False
    max_head = False
    mask_sigmoid = False

    sq_init = -2.5
    sx_init = 0.0
    l1_trans_loss = True


class I2PNetConfigNuScenes(I2PNetConfig):
    dataset_type = 1
    stride_Hs = [2, 2, 2, 1]
    init_H = 21
    fup = 2.0
    fdown = -24.8


CONFIGS = {
    "config_proj_lidarcenter": I2PNetConfig,
    "config_proj_lidarcenter_nus": I2PNetConfigNuScenes,
}
