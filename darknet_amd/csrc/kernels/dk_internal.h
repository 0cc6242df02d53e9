// dk_internal.h -- library-internal C++ entry points (not exported).
#pragma once
#include "dk_kernels.h"

// dk_conv_forward with an explicit tile configuration (cfg < 0: heuristic).
int dk_conv_forward_cfg(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream,
    int cfg);
// Makes sure the per-shape tap table exists (must happen outside stream capture).
void dk_conv_prepare(const DkConvDesc* d);
int dk_conv_num_configs();
// true when tile configuration `cfg` can run this layer (the direct 3x3 configurations,
// indices >= the number of gather configurations, only take 3x3/s1/p1 layers)
bool dk_conv_config_applicable(const DkConvDesc* d, int cfg);
