// resize.cpp -- ResizeNetwork: change the input resolution of a loaded network (multi-scale
// training, callers that switch the input size).  Reference twins (Ravicmoon/darknet src/):
// ResizeNetwork network.cpp:255-410 and the per-layer hooks it calls --
// resize_convolutional_layer convolutional_layer.cpp:805-914, ResizeMaxpoolLayer
// maxpool_layer.cpp:122-160, ResizeRouteLayer route_layer.c:46-85, ResizeShortcutLayer
// shortcut_layer.c:100-143, ResizeUpsampleLayer upsample_layer.c:49-74, ResizeYoloLayer
// yolo_layer.cpp:88-137, ResizeBatchnormLayer batchnorm_layer.cpp:166-204, ResizeAvgpoolLayer
// avgpool_layer.cpp:33-38, ResizeScaleChannelsLayer scale_channels_layer.c:50-68.
//
// Every layer re-derives its geometry from its inputs and re-allocates its device tensors; the
// train-mode delta arena, the conv tap tables and (for planned inference nets) the fusion /
// zero-copy / autotune plan are rebuilt afterwards, and a captured hipGraph is dropped.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"
#include "dk_internal.h"

void DkConvPrepare(layer* l);
void DkBuildDeltaArena(Network* net);

static void redev(float** p, size_t n)
{
  if (!*p)
    return;
  cuda_free(*p);
  *p = cuda_make_array(nullptr, n);
}

static void resize_layer_buffers(layer* l, bool has_indexes)
{
  const size_t total = (size_t)l->outputs * l->batch;
  if (!dk_gpu_enabled())
    return;
  redev(&l->output_gpu, total);
  if (l->delta_gpu)
  {
    if (!l->delta_in_arena)
      cuda_free(l->delta_gpu);
    l->delta_gpu = cuda_make_array(nullptr, total);   // moved into the new arena afterwards
    l->delta_in_arena = 0;
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total * sizeof(float), get_cuda_stream()));
  }
  redev(&l->x_gpu, total);
  redev(&l->x_norm_gpu, total);
  redev(&l->activation_input_gpu, total);
  if (has_indexes && l->indexes_gpu)
  {
    cuda_free((float*)l->indexes_gpu);
    l->indexes_gpu = cuda_make_int_array(total);
  }
}

void ResizeNetwork(Network* net, int w, int h)
{
  if (net->gpu_index >= 0)
  {
    cuda_set_device(net->gpu_index);
    NetworkSync(net);
  }
  DkInvalidateGraph(net);
  const bool gpu = net->gpu_index >= 0 && dk_gpu_enabled();
  // the arena owns every delta tensor: release it, the layers get fresh ones below
  if (gpu && net->delta_arena_gpu)
  {
    for (int i = 0; i < net->n; ++i)
      if (net->layers[i].delta_in_arena)
      {
        net->layers[i].delta_gpu = cuda_make_array(nullptr, (size_t)net->layers[i].outputs * net->layers[i].batch);  // own allocation again
        net->layers[i].delta_in_arena = 0;
      }
    cuda_free(net->delta_arena_gpu);
    net->delta_arena_gpu = nullptr;
    net->delta_arena_size = 0;
    net->delta_arena_zero = 0;
  }
  net->w = w;
  net->h = h;
  net->inputs = net->h * net->w * net->c;
  int inputs = net->inputs;
  size_t workspace_size = 0;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    // plan state refers to the old buffers
    l->out_view = nullptr;
    l->out_view_ctot = 0;
    l->out_alias = nullptr;
    l->conv_cfg = -1;
    l->train_plan[0] = l->train_plan[1] = l->train_plan[2] = 0;
    DkFreeTrainPrep(net);   // the kernel choices behind the derived-weights plan are void
    switch (l->type)
    {
      case CONVOLUTIONAL:
        l->w = w; l->h = h;
        l->out_w = (l->w + 2 * l->pad - l->size) / l->stride_x + 1;
        l->out_h = (l->h + 2 * l->pad - l->size) / l->stride_y + 1;
        l->outputs = l->out_h * l->out_w * l->out_c;
        l->inputs = l->w * l->h * l->c;
        resize_layer_buffers(l, false);
        l->workspace_size = (size_t)l->out_h * l->out_w * l->size * l->size * (l->c / l->groups) * sizeof(float);
        l->bflops = (2.0 * l->nweights * l->out_h * l->out_w) / 1000000000.;
        break;
      case MAXPOOL:
        l->h = h; l->w = w;
        l->inputs = h * w * l->c;
        l->out_w = (w + l->pad - l->size) / l->stride_x + 1;
        l->out_h = (h + l->pad - l->size) / l->stride_y + 1;
        l->outputs = l->out_w * l->out_h * l->out_c;
        resize_layer_buffers(l, true);
        break;
      case BATCHNORM:
        l->out_h = l->h = h;
        l->out_w = l->w = w;
        l->outputs = l->inputs = h * w * l->c;
        resize_layer_buffers(l, false);
        break;
      case AVGPOOL:
        l->w = w; l->h = h;
        l->inputs = h * w * l->c;
        break;
      case ROUTE:
      {
        layer* first = &net->layers[l->input_layers[0]];
        l->out_w = first->out_w;
        l->out_h = first->out_h;
        l->out_c = first->out_c;
        l->outputs = first->outputs;
        l->input_sizes[0] = first->outputs;
        for (int k = 1; k < l->n; ++k)
        {
          layer* next = &net->layers[l->input_layers[k]];
          l->outputs += next->outputs;
          l->input_sizes[k] = next->outputs;
          if (next->out_w == first->out_w && next->out_h == first->out_h)
            l->out_c += next->out_c;
          else
          {
            printf("Error: Different size of input layers: %d x %d, %d x %d\n", next->out_w, next->out_h, first->out_w, first->out_h);
            l->out_h = l->out_w = l->out_c = 0;
            exit(EXIT_FAILURE);
          }
        }
        l->out_c = l->out_c / l->groups;
        l->outputs = l->outputs / l->groups;
        l->inputs = l->outputs;
        l->w = first->w; l->h = first->h; l->c = l->out_c;
        resize_layer_buffers(l, false);
        break;
      }
      case SHORTCUT:
        l->w = l->out_w = w;
        l->h = l->out_h = h;
        l->outputs = w * h * l->out_c;
        l->inputs = l->outputs;
        l->input_sizes[0] = net->layers[l->index].outputs;
        resize_layer_buffers(l, false);
        break;
      case SCALE_CHANNELS:
      {
        layer* first = &net->layers[l->index];
        l->w = w; l->h = h;
        l->out_w = first->out_w;
        l->out_h = first->out_h;
        l->outputs = l->out_w * l->out_h * l->out_c;
        l->inputs = l->outputs;
        resize_layer_buffers(l, false);
        break;
      }
      case DROPOUT:
        l->inputs = l->outputs = inputs;
        l->out_w = l->w = w;
        l->out_h = l->h = h;
        l->output_gpu = net->layers[i - 1].output_gpu;
        l->delta_gpu = net->layers[i - 1].delta_gpu;
        // the train-mode mask is sized inputs*batch and allocated lazily by the forward slot
        // (resize_dropout_layer, dropout_layer.c:75-76, reallocates it): drop the old one
        if (gpu && l->rand_gpu)
        {
          cuda_free(l->rand_gpu);
          l->rand_gpu = nullptr;
        }
        break;
      case UPSAMPLE:
        l->w = w; l->h = h;
        l->out_w = w * l->stride;
        l->out_h = h * l->stride;
        l->outputs = l->out_w * l->out_h * l->out_c;
        l->inputs = l->h * l->w * l->c;
        resize_layer_buffers(l, false);
        break;
      case YOLO:
      case GAUSSIAN_YOLO:
      {
        l->w = w; l->h = h;
        l->out_w = w; l->out_h = h;
        l->outputs = h * w * l->n * (l->classes + (l->type == GAUSSIAN_YOLO ? 8 : 4) + 1);
        l->inputs = l->outputs;
        const size_t total = (size_t)l->batch * l->outputs;
        if (gpu)
        {
          if (l->output_pinned && l->output)
            cuda_free_host(l->output);
          l->output = cuda_make_array_pinned(nullptr, total);
          l->output_pinned = 1;
          memset(l->output, 0, total * sizeof(float));
          if (l->delta)
          {
            if (l->delta_pinned)
              cuda_free_host(l->delta);
            else
              free(l->delta);
            l->delta = cuda_make_array_pinned(nullptr, total);
            l->delta_pinned = 1;
          }
        }
        else
        {
          free(l->output);
          l->output = (float*)xcalloc(total, sizeof(float));
        }
        resize_layer_buffers(l, false);
        break;
      }
      default:
        fprintf(stderr, "Resizing type %d \n", (int)l->type);
        error("Cannot resize this type of layer");
    }
    if (l->workspace_size > workspace_size)
      workspace_size = l->workspace_size;
    inputs = l->outputs;
    w = l->out_w;
    h = l->out_h;
  }
  net->outputs = GetNetworkOutputSize(net);
  if (net->layers[net->n - 1].type == YOLO || net->layers[net->n - 1].type == GAUSSIAN_YOLO)
    net->output = net->layers[net->n - 1].output;
  if (gpu)
  {
    const size_t size = (size_t)GetNetworkInputSize(net) * net->batch;
    cuda_free(net->workspace);
    net->workspace = cuda_make_array(0, workspace_size / sizeof(float) + 1);
    cuda_free(net->input_state_gpu);
    net->input_state_gpu = cuda_make_array(0, size);
    if (net->input_pinned_cpu_flag && net->input_pinned_cpu)
      cuda_free_host(net->input_pinned_cpu);
    net->input_pinned_cpu = cuda_make_array_pinned(nullptr, size);
    net->input_pinned_cpu_flag = 1;
    if (net->train)
      DkBuildDeltaArena(net);
    for (int i = 0; i < net->n; ++i)
      if (net->layers[i].type == CONVOLUTIONAL)
        DkConvPrepare(&net->layers[i]);
    net->cand_valid = 0;
    net->f32_staged = 0;   // a float batch staged for the old resolution is dropped (DkNetworkStageFloat re-allocates)
    CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
    // the device-NMS head table holds the yolo grid sizes: rebuilt on the next DkGetNetworkBoxesNms
    if (net->nms_heads_gpu)
    {
      (void)hipFree(net->nms_heads_gpu);
      net->nms_heads_gpu = nullptr;
    }
    if (net->planned)
      DkPlanInference(net);
  }
}
