// Two builds of the same kernel sources share libru3d.so: the bf16 build owns the public C ABI names and forwards
// calls whose storage dtype is RU3D_F16 to the fp16 build (compiled with -DRU3D_STORAGE_F16), whose entry points carry
// the suffix _f16.  This header renames them there (before ru3d.h declares them) and declares them here.
#pragma once
#define RU3D_F16_APIS(X) \
    X(ru3d_packed_weight_bytes) \
    X(ru3d_pack_weight) \
    X(ru3d_pack_weights) \
    X(ru3d_conv3d_workspace_bytes) \
    X(ru3d_conv3d_fwd) \
    X(ru3d_conv3d_fwd_in_workspace_bytes) \
    X(ru3d_conv3d_fwd_in) \
    X(ru3d_conv3d_fwd_in_lrelu) \
    X(ru3d_planar_concat_supported) \
    X(ru3d_conv3d_dgrad) \
    X(ru3d_conv3d_s2_pair_fwd_in_supported) \
    X(ru3d_conv3d_s2_pair_fwd_in_workspace_bytes) \
    X(ru3d_conv3d_s2_pair_fwd_in) \
    X(ru3d_conv3d_s1_dgrad_pair_supported) \
    X(ru3d_conv3d_s1_dgrad_pair) \
    X(ru3d_conv3d_s2_dgrad_pair_supported) \
    X(ru3d_conv3d_s2_dgrad_pair) \
    X(ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes) \
    X(ru3d_convtranspose3d_k3s2p1_fwd_in) \
    X(ru3d_conv3d_dgrad_in_bwd_workspace_bytes) \
    X(ru3d_conv3d_dgrad_in_bwd) \
    X(ru3d_in_lrelu_bwd_apply) \
    X(ru3d_conv3d_wgrad_workspace_bytes) \
    X(ru3d_conv3d_wgrad) \
    X(ru3d_conv3d_wgrad_bias_workspace_bytes) \
    X(ru3d_conv3d_wgrad_bias) \
    X(ru3d_conv3d_wgrad_pair_supported) \
    X(ru3d_conv3d_wgrad_pair_workspace_bytes) \
    X(ru3d_conv3d_wgrad_pair) \
    X(ru3d_head_bwd_supported) \
    X(ru3d_head_bwd_workspace_bytes) \
    X(ru3d_head_bwd) \
    X(ru3d_convtranspose3d_k3s2p1_fwd) \
    X(ru3d_convtranspose3d_k3s2p1_dgrad) \
    X(ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes) \
    X(ru3d_convtranspose3d_k3s2p1_wgrad) \
    X(ru3d_instnorm_stats) \
    X(ru3d_in_lrelu_fwd) \
    X(ru3d_skip1x1_in_lrelu_fwd_supported) \
    X(ru3d_skip1x1_in_lrelu_fwd) \
    X(ru3d_in_lrelu_bwd) \
    X(ru3d_channel_sum) \
    X(ru3d_copy_channels) \
    X(ru3d_add) \
    X(ru3d_cast_f32) \
    X(ru3d_ncdhw_to_ndhwc) \
    X(ru3d_ndhwc_to_ncdhw) \
    X(ru3d_pointwise) \
    X(ru3d_batchnorm_stats_pool) \
    X(ru3d_affine_lrelu_fwd) \
    X(ru3d_batchnorm_bwd_pool) \
    X(ru3d_batchnorm_bwd_apply)

#ifdef RU3D_STORAGE_F16
#define ru3d_packed_weight_bytes ru3d_packed_weight_bytes_f16
#define ru3d_pack_weight ru3d_pack_weight_f16
#define ru3d_pack_weights ru3d_pack_weights_f16
#define ru3d_conv3d_workspace_bytes ru3d_conv3d_workspace_bytes_f16
#define ru3d_conv3d_fwd ru3d_conv3d_fwd_f16
#define ru3d_conv3d_fwd_in_workspace_bytes ru3d_conv3d_fwd_in_workspace_bytes_f16
#define ru3d_conv3d_fwd_in ru3d_conv3d_fwd_in_f16
#define ru3d_conv3d_fwd_in_lrelu ru3d_conv3d_fwd_in_lrelu_f16
#define ru3d_planar_concat_supported ru3d_planar_concat_supported_f16
#define ru3d_conv3d_dgrad ru3d_conv3d_dgrad_f16
#define ru3d_conv3d_s2_pair_fwd_in_supported ru3d_conv3d_s2_pair_fwd_in_supported_f16
#define ru3d_conv3d_s2_pair_fwd_in_workspace_bytes ru3d_conv3d_s2_pair_fwd_in_workspace_bytes_f16
#define ru3d_conv3d_s2_pair_fwd_in ru3d_conv3d_s2_pair_fwd_in_f16
#define ru3d_conv3d_s1_dgrad_pair_supported ru3d_conv3d_s1_dgrad_pair_supported_f16
#define ru3d_conv3d_s1_dgrad_pair ru3d_conv3d_s1_dgrad_pair_f16
#define ru3d_conv3d_s2_dgrad_pair_supported ru3d_conv3d_s2_dgrad_pair_supported_f16
#define ru3d_conv3d_s2_dgrad_pair ru3d_conv3d_s2_dgrad_pair_f16
#define ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes ru3d_convtranspose3d_k3s2p1_fwd_in_workspace_bytes_f16
#define ru3d_convtranspose3d_k3s2p1_fwd_in ru3d_convtranspose3d_k3s2p1_fwd_in_f16
#define ru3d_conv3d_dgrad_in_bwd_workspace_bytes ru3d_conv3d_dgrad_in_bwd_workspace_bytes_f16
#define ru3d_conv3d_dgrad_in_bwd ru3d_conv3d_dgrad_in_bwd_f16
#define ru3d_in_lrelu_bwd_apply ru3d_in_lrelu_bwd_apply_f16
#define ru3d_conv3d_wgrad_workspace_bytes ru3d_conv3d_wgrad_workspace_bytes_f16
#define ru3d_conv3d_wgrad ru3d_conv3d_wgrad_f16
#define ru3d_conv3d_wgrad_bias_workspace_bytes ru3d_conv3d_wgrad_bias_workspace_bytes_f16
#define ru3d_conv3d_wgrad_bias ru3d_conv3d_wgrad_bias_f16
#define ru3d_conv3d_wgrad_pair_supported ru3d_conv3d_wgrad_pair_supported_f16
#define ru3d_conv3d_wgrad_pair_workspace_bytes ru3d_conv3d_wgrad_pair_workspace_bytes_f16
#define ru3d_conv3d_wgrad_pair ru3d_conv3d_wgrad_pair_f16
#define ru3d_head_bwd_supported ru3d_head_bwd_supported_f16
#define ru3d_head_bwd_workspace_bytes ru3d_head_bwd_workspace_bytes_f16
#define ru3d_head_bwd ru3d_head_bwd_f16
#define ru3d_convtranspose3d_k3s2p1_fwd ru3d_convtranspose3d_k3s2p1_fwd_f16
#define ru3d_convtranspose3d_k3s2p1_dgrad ru3d_convtranspose3d_k3s2p1_dgrad_f16
#define ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes ru3d_convtranspose3d_k3s2p1_wgrad_workspace_bytes_f16
#define ru3d_convtranspose3d_k3s2p1_wgrad ru3d_convtranspose3d_k3s2p1_wgrad_f16
#define ru3d_instnorm_stats ru3d_instnorm_stats_f16
#define ru3d_in_lrelu_fwd ru3d_in_lrelu_fwd_f16
#define ru3d_skip1x1_in_lrelu_fwd_supported ru3d_skip1x1_in_lrelu_fwd_supported_f16
#define ru3d_skip1x1_in_lrelu_fwd ru3d_skip1x1_in_lrelu_fwd_f16
#define ru3d_in_lrelu_bwd ru3d_in_lrelu_bwd_f16
#define ru3d_channel_sum ru3d_channel_sum_f16
#define ru3d_copy_channels ru3d_copy_channels_f16
#define ru3d_add ru3d_add_f16
#define ru3d_cast_f32 ru3d_cast_f32_f16
#define ru3d_ncdhw_to_ndhwc ru3d_ncdhw_to_ndhwc_f16
#define ru3d_ndhwc_to_ncdhw ru3d_ndhwc_to_ncdhw_f16
#define ru3d_pointwise ru3d_pointwise_f16
#define ru3d_batchnorm_stats_pool ru3d_batchnorm_stats_pool_f16
#define ru3d_affine_lrelu_fwd ru3d_affine_lrelu_fwd_f16
#define ru3d_batchnorm_bwd_pool ru3d_batchnorm_bwd_pool_f16
#define ru3d_batchnorm_bwd_apply ru3d_batchnorm_bwd_apply_f16
#endif
