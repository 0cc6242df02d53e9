// train.cpp -- training-mode layer paths (BN with batch statistics, backward,
// update).  Round-1 status: the inference path (BN folded) is complete; the
// train-mode conv forward below is the un-fused sequence conv GEMM -> BN
// (rolling statistics when state.train == 0) -> activation.
#include <stdio.h>
#include <stdlib.h>

#include "dk_host.h"
#include "dk_internal.h"

void ForwardConvTrainGpu(layer* l, NetworkState state)
{
  (void)state;
  fprintf(stderr,
      "darknet_amd: layer %d: convolution with un-folded batch_normalize (train-mode load) is not "
      "implemented in this build; load with LoadNetwork(train=false) / LoadNetworkBatch.\n",
      l->index);
  exit(EXIT_FAILURE);
}
