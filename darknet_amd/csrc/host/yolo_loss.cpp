// yolo_loss.cpp -- the YOLO training delta / cost, host C++ exactly like in the
// reference (even its GPU build computes this on the host after a D2H pull and
// pushes the delta back: src/yolo_layer.cpp:861-881).  SURVEY.md section 8 row a8b.
//
// Own restatement of ForwardYoloLayer's train branch (src/yolo_layer.cpp:413-772)
// with its helpers delta_yolo_box :172-273, delta_yolo_class :295-362,
// averages_yolo_deltas :275-293, compare_yolo_class :364-378, GetYoloBox :139-148
// and the IoU family + analytic IoU-loss gradient of src/box.cpp:36-351.  The
// float/double evaluation of every expression follows the reference's C++ build
// (float overloads of exp/log/atan/pow on float arguments, double where a double
// literal or M_PI takes part), because the result is pinned BIT-EXACTLY against
// golden vectors dumped from the real reference (tests/golden/yololoss_*.npz).
#include <float.h>
#include <math.h>
#include <algorithm>
#include <atomic>
#include <thread>
#include <utility>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"

namespace
{
inline float sq(float v) { return v * v; }

struct Edges
{
  float left, right, top, bottom;
};

inline Edges edges_of(const Box& b)
{
  return Edges{b.x - b.w / 2.0f, b.x + b.w / 2.0f, b.y - b.h / 2.0f, b.y + b.h / 2.0f};
}

// GetYoloBox, yolo_layer.cpp:139-148
inline Box yolo_box(const float* x, const float* biases, int a, int index, int col, int row, int lw,
    int lh, int net_w, int net_h, int stride)
{
  Box b;
  b.x = (col + x[index + 0 * stride]) / lw;
  b.y = (row + x[index + 1 * stride]) / lh;
  b.w = expf(x[index + 2 * stride]) * biases[2 * a] / net_w;
  b.h = expf(x[index + 3 * stride]) * biases[2 * a + 1] / net_h;
  return b;
}

// Box::Giou, box.cpp:115-131
float giou_of(const Box& p, const Box& t)
{
  const Edges a = edges_of(p), b = edges_of(t);
  const float w = fmaxf(a.right, b.right) - fminf(a.left, b.left);
  const float h = fmaxf(a.bottom, b.bottom) - fminf(a.top, b.top);
  const float c = w * h;
  const float iou = Box::Iou(p, t);
  if (fabsf(c) < FLT_EPSILON)
    return iou;
  const float u = Box::Union(p, t);
  return iou - (c - u) / c;
}

// Box::Ciou, box.cpp:73-96 (needed only as an anchor-matching metric: iou_thresh_kind)
float ciou_of(const Box& p, const Box& t)
{
  const Edges a = edges_of(p), b = edges_of(t);
  const float w = fmaxf(a.right, b.right) - fminf(a.left, b.left);
  const float h = fmaxf(a.bottom, b.bottom) - fminf(a.top, b.top);
  const float c = w * w + h * h;
  const float iou = Box::Iou(p, t);
  if (fabsf(c) < FLT_EPSILON)
    return iou;
  const float u = sq(p.x - t.x) + sq(p.y - t.y);
  const float d = u / c;
  const float ar_gt = t.w / t.h, ar_pred = p.w / p.h;
  const float ar_loss = 4 / (M_PI * M_PI) * sq(atanf(ar_gt) - atanf(ar_pred));
  const float alpha = ar_loss / (1 - iou + ar_loss + 0.000001);
  return iou - (d + alpha * ar_loss);
}

float rmse_of(const Box& a, const Box& b)
{
  return sqrtf(sq(a.x - b.x) + sq(a.y - b.y) + sq(a.w - b.w) + sq(a.h - b.h));
}

// Box::Iou(b1, b2, kind), box.cpp:133-151
float iou_kind(const Box& a, const Box& b, IOU_LOSS kind)
{
  switch (kind)
  {
    case GIOU: return giou_of(a, b);
    case MSE: return rmse_of(a, b);
    case DIOU: return Box::Diou(a, b);
    case CIOU: return ciou_of(a, b);
    default: return Box::Iou(a, b);
  }
}

struct BoxGrad
{
  float dx, dy, dw, dh;
};

// Box::DxIou, box.cpp:153-351: gradient of IoU / GIoU / DIoU / CIoU w.r.t. the
// predicted box (x, y, w, h).
BoxGrad iou_gradient(const Box& pred, const Box& gt, IOU_LOSS kind)
{
  const Edges pe = edges_of(pred), ge = edges_of(gt);
  const float pt = fminf(pe.top, pe.bottom), pb = fmaxf(pe.top, pe.bottom);
  const float pl = fminf(pe.left, pe.right), pr = fmaxf(pe.left, pe.right);

  const float area_p = (pb - pt) * (pr - pl);
  const float area_g = (ge.bottom - ge.top) * (ge.right - ge.left);
  const float ih = fminf(pb, ge.bottom) - fmaxf(pt, ge.top);
  const float iw = fminf(pr, ge.right) - fmaxf(pl, ge.left);
  const float inter = iw * ih;
  const float uni = area_p + area_g - inter;
  const float dist2 = sq(pred.x - gt.x) + sq(pred.y - gt.y);
  const float enc_w = fmaxf(pr, ge.right) - fminf(pl, ge.left);
  const float enc_h = fmaxf(pb, ge.bottom) - fminf(pt, ge.top);
  const float enc_area = enc_w * enc_h;

  // d(area_p)/d(edge), d(inter)/d(edge), d(union)/d(edge), d(enclosing)/d(edge); order t,b,l,r
  const float dA[4] = {-1 * (pr - pl), pr - pl, -1 * (pb - pt), pb - pt};
  const float dI[4] = {pt > ge.top ? (-1 * iw) : 0, pb < ge.bottom ? iw : 0,
      pl > ge.left ? (-1 * ih) : 0, pr < ge.right ? ih : 0};
  float dU[4], dC[4] = {pt < ge.top ? (-1 * enc_w) : 0, pb > ge.bottom ? enc_w : 0,
                   pl < ge.left ? (-1 * enc_h) : 0, pr > ge.right ? enc_h : 0};
  for (int e = 0; e < 4; ++e) dU[e] = dA[e] - dI[e];

  float g[4] = {0, 0, 0, 0};  // t, b, l, r
  if (uni > 0)
    for (int e = 0; e < 4; ++e) g[e] = ((uni * dI[e]) - (inter * dU[e])) / (uni * uni);
  // corner selection for degenerate (flipped) predictions, in the reference's order
  g[0] = pe.top < pe.bottom ? g[0] : g[1];
  g[1] = pe.top < pe.bottom ? g[1] : g[0];
  g[2] = pe.left < pe.right ? g[2] : g[3];
  g[3] = pe.left < pe.right ? g[3] : g[2];

  if (kind == GIOU)
  {
    if (enc_area > 0)
      for (int e = 0; e < 4; ++e)
        g[e] += ((enc_area * dU[e]) - (uni * dC[e])) / (enc_area * enc_area);
    if (iw <= 0 || ih <= 0)
      for (int e = 0; e < 4; ++e)
        g[e] = ((enc_area * dU[e]) - (uni * dC[e])) / (enc_area * enc_area);
  }

  // centre-distance penalty terms (DIoU / CIoU)
  const float ct = fminf(pred.y - pred.h / 2, gt.y - gt.h / 2);
  const float cb = fmaxf(pred.y + pred.h / 2, gt.y + gt.h / 2);
  const float cl = fminf(pred.x - pred.w / 2, gt.x - gt.w / 2);
  const float cr = fmaxf(pred.x + pred.w / 2, gt.x + gt.w / 2);
  const float cw = cr - cl, ch = cb - ct;
  const float diag2 = sq(cw) + sq(ch);

  const float dct_dy = pt < ge.top ? 1 : 0, dct_dh = pt < ge.top ? -0.5 : 0;
  const float dcb_dy = pb > ge.bottom ? 1 : 0, dcb_dh = pb > ge.bottom ? 0.5 : 0;
  const float dcl_dx = pl < ge.left ? 1 : 0, dcl_dw = pl < ge.left ? -0.5 : 0;
  const float dcr_dx = pr > ge.right ? 1 : 0, dcr_dw = pr > ge.right ? 0.5 : 0;
  const float dcw_dx = dcr_dx - dcl_dx, dcw_dy = 0.f - 0.f, dcw_dw = dcr_dw - dcl_dw, dcw_dh = 0.f - 0.f;
  const float dch_dx = 0.f - 0.f, dch_dy = dcb_dy - dct_dy, dch_dw = 0.f - 0.f, dch_dh = dcb_dh - dct_dh;

  BoxGrad r;
  r.dx = g[2] + g[3];
  r.dy = g[0] + g[1];
  r.dw = (g[3] - g[2]);
  r.dh = (g[1] - g[0]);

  auto pen_x = [&]() { return (2 * (gt.x - pred.x) * diag2 - (2 * cw * dcw_dx + 2 * ch * dch_dx) * dist2) / sq(diag2); };
  auto pen_y = [&]() { return (2 * (gt.y - pred.y) * diag2 - (2 * cw * dcw_dy + 2 * ch * dch_dy) * dist2) / sq(diag2); };
  auto pen_w = [&]() { return (2 * cw * dcw_dw + 2 * ch * dch_dw) * dist2 / sq(diag2); };
  auto pen_h = [&]() { return (2 * cw * dcw_dh + 2 * ch * dch_dh) * dist2 / sq(diag2); };

  if (kind == DIOU)
  {
    if (diag2 > 0)
    {
      r.dx += pen_x(); r.dy += pen_y(); r.dw += pen_w(); r.dh += pen_h();
    }
    if (iw <= 0 || ih <= 0)
    {
      r.dx = pen_x(); r.dy = pen_y(); r.dw = pen_w(); r.dh = pen_h();
    }
  }
  if (kind == CIOU)
  {
    const float ar_gt = gt.w / gt.h, ar_pred = pred.w / pred.h;
    const float ar_loss = 4 / (M_PI * M_PI) * sq(atanf(ar_gt) - atanf(ar_pred));
    const float alpha = ar_loss / (1 - inter / uni + ar_loss + 0.000001);
    const float ar_dw = 8 / (M_PI * M_PI) * (atanf(ar_gt) - atanf(ar_pred)) * pred.h;
    const float ar_dh = -8 / (M_PI * M_PI) * (atanf(ar_gt) - atanf(ar_pred)) * pred.w;
    if (diag2 > 0)
    {
      r.dx += pen_x();
      r.dy += pen_y();
      r.dw += pen_w() + alpha * ar_dw;
      r.dh += pen_h() + alpha * ar_dh;
    }
    if (iw <= 0 || ih <= 0)
    {
      r.dx = pen_x();
      r.dy = pen_y();
      r.dw = pen_w() + alpha * ar_dw;
      r.dh = pen_h() + alpha * ar_dh;
    }
  }
  return r;
}

inline float zero_if_not_finite(float v) { return (isnan(v) || isinf(v)) ? 0 : v; }
inline float clamp_abs(float v, float m) { return v > m ? m : (v < -m ? -m : v); }

struct IouPair
{
  float iou, giou;
};

// delta_yolo_box, yolo_layer.cpp:172-273 (accumulate = 1 at every call site)
IouPair box_delta(const Box& truth, const float* x, const float* biases, int a, int index, int col,
    int row, int lw, int lh, int net_w, int net_h, float* delta, float scale, int stride,
    float iou_normalizer, IOU_LOSS iou_loss, float max_delta)
{
  Box pred = yolo_box(x, biases, a, index, col, row, lw, lh, net_w, net_h, stride);
  IouPair r;
  r.iou = Box::Iou(pred, truth);
  r.giou = giou_of(pred, truth);
  if (pred.w == 0)
    pred.w = 1.0;
  if (pred.h == 0)
    pred.h = 1.0;
  if (iou_loss == MSE)
  {
    const float tx = (truth.x * lw - col);
    const float ty = (truth.y * lh - row);
    const float tw = logf(truth.w * net_w / biases[2 * a]);
    const float th = logf(truth.h * net_h / biases[2 * a + 1]);
    delta[index + 0 * stride] += scale * (tx - x[index + 0 * stride]) * iou_normalizer;
    delta[index + 1 * stride] += scale * (ty - x[index + 1 * stride]) * iou_normalizer;
    delta[index + 2 * stride] += scale * (tw - x[index + 2 * stride]) * iou_normalizer;
    delta[index + 3 * stride] += scale * (th - x[index + 3 * stride]) * iou_normalizer;
    return r;
  }
  const BoxGrad g = iou_gradient(pred, truth, iou_loss);
  float dx = g.dx, dy = g.dy, dw = g.dw, dh = g.dh;
  dw *= expf(x[index + 2 * stride]);  // chain rule through w = exp(tw)
  dh *= expf(x[index + 3 * stride]);
  dx *= iou_normalizer; dy *= iou_normalizer; dw *= iou_normalizer; dh *= iou_normalizer;
  dx = zero_if_not_finite(dx); dy = zero_if_not_finite(dy);
  dw = zero_if_not_finite(dw); dh = zero_if_not_finite(dh);
  if (max_delta != FLT_MAX)
  {
    dx = clamp_abs(dx, max_delta); dy = clamp_abs(dy, max_delta);
    dw = clamp_abs(dw, max_delta); dh = clamp_abs(dh, max_delta);
  }
  delta[index + 0 * stride] += dx;
  delta[index + 1 * stride] += dy;
  delta[index + 2 * stride] += dw;
  delta[index + 3 * stride] += dh;
  return r;
}

// delta_yolo_class, yolo_layer.cpp:295-362
void class_delta(const float* output, float* delta, int index, int class_id, int classes, int stride,
    float* avg_cat, int focal_loss, float label_smooth_eps, const float* multipliers)
{
  if (delta[index + stride * class_id])
  {
    float y_true = 1;
    if (label_smooth_eps)
      y_true = y_true * (1 - label_smooth_eps) + 0.5 * label_smooth_eps;
    const float d = y_true - output[index + stride * class_id];
    if (!isnan(d) && !isinf(d))
      delta[index + stride * class_id] = d;
    if (multipliers)
      delta[index + stride * class_id] *= multipliers[class_id];
    if (avg_cat)
      *avg_cat += output[index + stride * class_id];
    return;
  }
  if (focal_loss)
  {
    const float alpha = 0.5;
    const float pt = output[index + stride * class_id] + 0.000000000000001F;
    const float grad = -(1 - pt) * (2 * pt * logf(pt) + pt - 1);
    for (int n = 0; n < classes; ++n)
    {
      delta[index + stride * n] = (((n == class_id) ? 1 : 0) - output[index + stride * n]);
      delta[index + stride * n] *= alpha * grad;
      if (n == class_id && avg_cat)
        *avg_cat += output[index + stride * n];
    }
    return;
  }
  for (int n = 0; n < classes; ++n)
  {
    float y_true = ((n == class_id) ? 1 : 0);
    if (label_smooth_eps)
      y_true = y_true * (1 - label_smooth_eps) + 0.5 * label_smooth_eps;
    const float d = y_true - output[index + stride * n];
    if (!isnan(d) && !isinf(d))
      delta[index + stride * n] = d;
    if (multipliers && n == class_id)
      delta[index + stride * class_id] *= multipliers[class_id];
    if (n == class_id && avg_cat)
      *avg_cat += output[index + stride * n];
  }
}

inline int entry(const layer* l, int b, int location, int e)
{
  const int n = location / (l->w * l->h);
  const int loc = location % (l->w * l->h);
  return b * l->outputs + n * l->w * l->h * (4 + l->classes + 1) + e * l->w * l->h + loc;
}

}  // namespace

// The train branch of ForwardYoloLayer on host arrays: `out` = decoded yolo output
// [batch][outputs] (NaN/Inf objectness is zeroed in place, as the reference does),
// `truth` = [batch][max_boxes*5], `delta` (zero-filled here) receives the loss
// gradient; returns *(l->cost).
namespace
{
struct ImageLoss
{
  std::vector<std::pair<float, float>> iou_terms;  // (1 - iou, 1 - giou) in the order the reference adds them
  std::vector<std::vector<int>> positives;          // per anchor: cells whose class/box deltas were written
};

// passes 1-3 of the train branch for ONE image (images are independent)
void yolo_loss_image(const layer* l, int net_w, int net_h, float* out, const float* truth,
    float* delta, int b, ImageLoss* res)
{
  const int stride = l->w * l->h;
  memset(delta + (size_t)b * l->outputs, 0, (size_t)l->outputs * sizeof(float));
  res->positives.assign(l->n, std::vector<int>());
  float avg_cat = 0;
  auto truth_box = [&](int bb, int t) {
    const float* f = truth + t * (4 + 1) + bb * l->truths;
    return Box(f[0], f[1], f[2], f[3]);
  };
  auto truth_class = [&](int bb, int t) { return (int)truth[t * (4 + 1) + bb * l->truths + 4]; };
  {
    // ---- pass 1: every predictor -> no-object delta unless it overlaps a truth well
    for (int j = 0; j < l->h; ++j)
      for (int i = 0; i < l->w; ++i)
        for (int n = 0; n < l->n; ++n)
        {
          const int loc = n * stride + j * l->w + i;
          const int box_index = entry(l, b, loc, 0);
          const int obj_index = entry(l, b, loc, 4);
          const int class_index = entry(l, b, loc, 4 + 1);
          const Box pred = yolo_box(out, l->biases, l->mask[n], box_index, i, j, l->w, l->h, net_w,
              net_h, stride);
          float best_match_iou = 0, best_iou = 0;
          int best_t = 0;
          int class_match = -1;  // evaluated once per cell (the reference re-evaluates it per truth)
          for (int t = 0; t < l->max_boxes; ++t)
          {
            const Box tb = truth_box(b, t);
            const int class_id = truth_class(b, t);
            if (class_id >= l->classes || class_id < 0)
            {
              printf("\n Warning: in txt-labels class_id=%d >= classes=%d in cfg-file. \n", class_id,
                  l->classes);
              continue;
            }
            if (!tb.x)
              break;
            const float objectness = out[obj_index];
            if (isnan(objectness) || isinf(objectness))
              out[obj_index] = 0;
            // compare_yolo_class: does ANY class probability exceed 0.25?
            if (class_match < 0)
            {
              class_match = 0;
              for (int c = 0; c < l->classes; ++c)
                if (out[class_index + stride * c] > 0.25f)
                {
                  class_match = 1;
                  break;
                }
            }
            const float iou = Box::Iou(pred, tb);
            if (iou > best_match_iou && class_match == 1)
              best_match_iou = iou;
            if (iou > best_iou)
            {
              best_iou = iou;
              best_t = t;
            }
          }
          delta[obj_index] = l->cls_normalizer * (0 - out[obj_index]);
          if (best_match_iou > l->ignore_thresh)
            delta[obj_index] = 0;
          if (best_iou > l->truth_thresh)
          {
            delta[obj_index] = l->cls_normalizer * (1 - out[obj_index]);
            int class_id = truth_class(b, best_t);
            if (l->map)
              class_id = l->map[class_id];
            class_delta(out, delta, class_index, class_id, l->classes, stride, 0, l->focal_loss,
                l->label_smooth_eps, l->classes_multipliers);
            const Box tb = truth_box(b, best_t);
            const float mult = l->classes_multipliers ? l->classes_multipliers[class_id] : 1.0f;
            box_delta(tb, out, l->biases, l->mask[n], box_index, i, j, l->w, l->h, net_w, net_h,
                delta, (2 - tb.w * tb.h), stride, l->iou_normalizer * mult, l->iou_loss, l->max_delta);
            res->positives[n].push_back(j * l->w + i);
          }
        }

    // ---- pass 2: every truth -> its best anchor (and the anchors above iou_thresh)
    for (int t = 0; t < l->max_boxes; ++t)
    {
      const Box tb = truth_box(b, t);
      if (tb.x < 0 || tb.y < 0 || tb.x > 1 || tb.y > 1 || tb.w < 0 || tb.h < 0)
        printf(" Wrong label: truth.x = %f, truth.y = %f, truth.w = %f, truth.h = %f \n", tb.x, tb.y,
            tb.w, tb.h);
      int class_id = truth_class(b, t);
      if (class_id >= l->classes || class_id < 0)
        continue;
      if (!tb.x)
        break;
      float best_iou = 0;
      int best_n = 0;
      const int i = (int)(tb.x * l->w);
      const int j = (int)(tb.y * l->h);
      Box shifted = tb;
      shifted.x = shifted.y = 0;
      for (int n = 0; n < l->total; ++n)
      {
        Box anchor;
        anchor.w = l->biases[2 * n] / net_w;
        anchor.h = l->biases[2 * n + 1] / net_h;
        const float iou = Box::Iou(anchor, shifted);
        if (iou > best_iou)
        {
          best_iou = iou;
          best_n = n;
        }
      }
      if (l->map)
        class_id = l->map[class_id];
      const float mult = l->classes_multipliers ? l->classes_multipliers[class_id] : 1.0f;
      auto assign = [&](int anchor, int mask_n) {
        const int loc = mask_n * stride + j * l->w + i;
        const int box_index = entry(l, b, loc, 0);
        const IouPair ious = box_delta(tb, out, l->biases, anchor, box_index, i, j, l->w, l->h,
            net_w, net_h, delta, (2 - tb.w * tb.h), stride, l->iou_normalizer * mult, l->iou_loss,
            l->max_delta);
        res->iou_terms.emplace_back(1 - ious.iou, 1 - ious.giou);
        const int obj_index = entry(l, b, loc, 4);
        delta[obj_index] = mult * l->cls_normalizer * (1 - out[obj_index]);
        const int class_index = entry(l, b, loc, 4 + 1);
        class_delta(out, delta, class_index, class_id, l->classes, stride, &avg_cat, l->focal_loss,
            l->label_smooth_eps, l->classes_multipliers);
        res->positives[mask_n].push_back(j * l->w + i);
      };
      auto mask_index = [&](int anchor) {
        for (int k = 0; k < l->n; ++k)
          if (l->mask[k] == anchor)
            return k;
        return -1;
      };
      const int best_mask = mask_index(best_n);
      if (best_mask >= 0)
        assign(best_n, best_mask);
      for (int n = 0; n < l->total; ++n)
      {
        const int mask_n = mask_index(n);
        if (mask_n >= 0 && n != best_n && l->iou_thresh < 1.0f)
        {
          Box anchor;
          anchor.w = l->biases[2 * n] / net_w;
          anchor.h = l->biases[2 * n + 1] / net_h;
          if (iou_kind(anchor, shifted, l->iou_thresh_kind) > l->iou_thresh)
            assign(n, mask_n);
        }
      }
    }

    // ---- pass 3: a box shared by several positive classes gets its delta averaged.
    // Class deltas are non-zero only at the cells recorded above (everything else
    // was zero-filled and never written), so only those cells can have positives.
    for (int n = 0; n < l->n; ++n)
    {
      std::vector<int>& cells = res->positives[n];
      std::sort(cells.begin(), cells.end());
      cells.erase(std::unique(cells.begin(), cells.end()), cells.end());
      for (int cell : cells)
      {
        const int loc = n * stride + cell;
        const int box_index = entry(l, b, loc, 0), class_index = entry(l, b, loc, 4 + 1);
        int positives = 0;
        for (int c = 0; c < l->classes; ++c)
          if (delta[class_index + stride * c] > 0)
            positives++;
        if (positives > 0)
          for (int e = 0; e < 4; ++e) delta[box_index + e * stride] /= positives;
      }
    }
  }
}
}  // namespace

// The train branch of ForwardYoloLayer on host arrays: `out` = decoded yolo output
// [batch][outputs] (NaN/Inf objectness is zeroed in place, as the reference does),
// `truth` = [batch][max_boxes*5], `delta` (zero-filled here) receives the loss
// gradient; returns *(l->cost).  Images are processed by parallel threads (they are
// independent); every floating-point accumulation that crosses images or cells is
// then replayed in the reference's order, so deltas AND cost are bit-identical to the
// sequential reference (tests/test_yolo_loss_cpu.py).
extern "C" LIB_API float DkYoloLossHost(const layer* l, int net_w, int net_h, float* out,
    const float* truth, float* delta)
{
  const int stride = l->w * l->h;
  std::vector<ImageLoss> img(l->batch);
  {
    unsigned hw = std::thread::hardware_concurrency();
    int nthreads = (int)(hw ? hw : 1);
    if (const char* e = getenv("DK_LOSS_THREADS"))
      nthreads = atoi(e);
    if (nthreads > l->batch) nthreads = l->batch;
    if (nthreads > 16) nthreads = 16;
    if (nthreads <= 1)
      for (int b = 0; b < l->batch; ++b) yolo_loss_image(l, net_w, net_h, out, truth, delta, b, &img[b]);
    else
    {
      std::atomic<int> next(0);
      std::vector<std::thread> pool;
      for (int t = 0; t < nthreads; ++t)
        pool.emplace_back([&]() {
          for (int b = next++; b < l->batch; b = next++)
            yolo_loss_image(l, net_w, net_h, out, truth, delta, b, &img[b]);
        });
      for (auto& th : pool) th.join();
    }
  }
  float tot_iou_loss = 0, tot_giou_loss = 0;
  int count = 0;
  for (int b = 0; b < l->batch; ++b)
    for (const auto& t : img[b].iou_terms)
    {
      tot_iou_loss += t.first;
      tot_giou_loss += t.second;
      ++count;
    }
  if (count == 0)
    count = 1;

  // cost: iou part (mean 1-IoU or 1-GIoU) + cls_normalizer * |delta without the box terms|^2.
  // The reference sums delta[i]^2 over the whole array in index order (mag_array); the
  // structurally zero entries (class/box deltas of cells never marked positive) add
  // exactly 0, so only the objectness planes and the positive cells are visited --
  // in the same index order, hence the same float result.
  auto sum_squares = [&](bool with_box) {
    float sum = 0;
    for (int b = 0; b < l->batch; ++b)
      for (int n = 0; n < l->n; ++n)
      {
        const std::vector<int>& cells = img[b].positives[n];
        for (int e = 0; e < 4 + 1 + l->classes; ++e)
        {
          const float* plane = delta + entry(l, b, n * stride, e);
          if (e < 4 && !with_box)
            continue;
          if (e == 4)
            for (int i = 0; i < stride; ++i) sum += plane[i] * plane[i];
          else
            for (int cell : cells) sum += plane[cell] * plane[cell];
        }
      }
    return sum;
  };
  const float m = sqrtf(sum_squares(false));
  const float classification_loss = l->cls_normalizer * pow(m, 2);
  float cost;
  if (l->iou_loss == MSE)
    cost = pow(sqrtf(sum_squares(true)), 2);
  else
  {
    const float avg_iou_loss = (l->iou_loss == GIOU) ? l->iou_normalizer * (tot_giou_loss / count)
                                                     : l->iou_normalizer * (tot_iou_loss / count);
    cost = avg_iou_loss + classification_loss;
  }
  return cost;
}


// ------------------------------------------------------------------------------------------------
// [Gaussian_yolo] training delta / cost: own restatement of ForwardGaussianYoloLayer's
// train branch (src/gaussian_yolo_layer.cpp:518-851) with delta_gaussian_yolo_box :195-405,
// AveragesGaussianYoloDeltas :407-428, DeltaGaussianYoloClass :430-461, CompareGaussianYoloClass :463-475 and
// GetGaussianYoloBox :151-177.  Host code in the reference too (its GPU build pulls the head, runs this and
// pushes the delta: :968-995).  Pinned BIT-EXACTLY against the real reference: tests/golden/gaussianloss.npz.
// counters_per_class and map= are not supported by the parser here (classes_multipliers == NULL, no map).
namespace
{
inline int gentry(const layer* l, int b, int location, int e)
{
  const int n = location / (l->w * l->h);
  const int loc = location % (l->w * l->h);
  return b * l->outputs + n * l->w * l->h * (8 + l->classes + 1) + e * l->w * l->h + loc;
}

inline Box gaussian_box(const float* x, const float* biases, int a, int index, int i, int j, int lw, int lh,
    int w, int h, int stride, YOLO_POINT yp)
{
  Box b;
  b.w = expf(x[index + 4 * stride]) * biases[2 * a] / w;
  b.h = expf(x[index + 6 * stride]) * biases[2 * a + 1] / h;
  b.x = (i + x[index + 0 * stride]) / lw;
  b.y = (j + x[index + 2 * stride]) / lh;
  if (yp == YOLO_LEFT_TOP)
  {
    b.x = (i + x[index + 0 * stride]) / lw + b.w / 2;
    b.y = (j + x[index + 2 * stride]) / lh + b.h / 2;
  }
  else if (yp == YOLO_RIGHT_BOTTOM)
  {
    b.x = (i + x[index + 0 * stride]) / lw - b.w / 2;
    b.y = (j + x[index + 2 * stride]) / lh - b.h / 2;
  }
  return b;
}

// one coordinate of the negative-log-likelihood gradient: (mu gradient, sigma gradient) for the residual d and
// the predicted sigma, in the reference's mixed float / double evaluation
struct NllGrad
{
  float mu, sigma;
};
inline NllGrad nll_grad(float d, float sigma, float scale)
{
  const float sigma_const = 0.3;
  const float epsi = pow(10, -9);
  const float in_exp = d / sigma;
  const float in_exp_2 = pow((double)in_exp, 2.0);   // std::pow(float, int) promotes to double
  const float normal_dist = exp(in_exp_2 * (-1. / 2.)) / (sqrt(M_PI * 2.0) * (sigma + sigma_const));
  const float temp = (1. / 2.) * 1. / (normal_dist + epsi) * normal_dist * scale;
  NllGrad g;
  g.mu = temp * in_exp * (1. / sigma);
  g.sigma = temp * (in_exp_2 / sigma - 1. / (sigma + sigma_const));
  return g;
}

// delta_gaussian_yolo_box (accumulate = 1 at every call site); returns the IoU it reports
float gaussian_box_delta(const Box& truth, const float* x, const float* biases, int a, int index, int i, int j, int lw,
    int lh, int w, int h, float* delta, float scale, int stride, float iou_normalizer, IOU_LOSS iou_loss,
    float uc_normalizer, YOLO_POINT yp, float max_delta)
{
  Box pred = gaussian_box(x, biases, a, index, i, j, lw, lh, w, h, stride, yp);
  float iou = Box::Iou(pred, truth);
  const float giou = giou_of(pred, truth);
  if (pred.w == 0)
    pred.w = 1.0;
  if (pred.h == 0)
    pred.h = 1.0;
  float tx = (truth.x * lw - i);
  float ty = (truth.y * lh - j);
  const float tw = logf(truth.w * w / biases[2 * a]);
  const float th = logf(truth.h * h / biases[2 * a + 1]);
  if (yp == YOLO_LEFT_TOP)
  {
    tx = ((truth.x - truth.w / 2) * lw - i);
    ty = ((truth.y - truth.h / 2) * lh - j);
  }
  else if (yp == YOLO_RIGHT_BOTTOM)
  {
    tx = ((truth.x + truth.w / 2) * lw - i);
    ty = ((truth.y + truth.h / 2) * lh - j);
  }
  const NllGrad gx = nll_grad(tx - x[index + 0 * stride], x[index + 1 * stride], scale);
  const NllGrad gy = nll_grad(ty - x[index + 2 * stride], x[index + 3 * stride], scale);
  const NllGrad gw = nll_grad(tw - x[index + 4 * stride], x[index + 5 * stride], scale);
  const NllGrad gh = nll_grad(th - x[index + 6 * stride], x[index + 7 * stride], scale);
  float d_mu[4] = {gx.mu, gy.mu, gw.mu, gh.mu};
  float d_sg[4] = {gx.sigma, gy.sigma, gw.sigma, gh.sigma};
  if (iou_loss != MSE)
  {
    iou = giou;
    const BoxGrad g = iou_gradient(pred, truth, iou_loss);
    float dx = g.dx, dy = g.dy, dw = g.dw, dh = g.dh;
    if (yp == YOLO_LEFT_TOP)
    {
      dx = dx - dw / 2;
      dy = dy - dh / 2;
    }
    else if (yp == YOLO_RIGHT_BOTTOM)
    {
      dx = dx + dw / 2;
      dy = dy + dh / 2;
    }
    dw *= expf(x[index + 4 * stride]);
    dh *= expf(x[index + 6 * stride]);
    d_mu[0] = dx; d_mu[1] = dy; d_mu[2] = dw; d_mu[3] = dh;
  }
  for (int k = 0; k < 4; ++k)
  {
    d_mu[k] *= iou_normalizer;
    d_sg[k] *= uc_normalizer;
  }
  for (int k = 0; k < 4; ++k)
  {
    d_mu[k] = zero_if_not_finite(d_mu[k]);
    d_sg[k] = zero_if_not_finite(d_sg[k]);
  }
  if (max_delta != FLT_MAX)
    for (int k = 0; k < 4; ++k)
    {
      d_mu[k] = clamp_abs(d_mu[k], max_delta);
      d_sg[k] = clamp_abs(d_sg[k], max_delta);
    }
  for (int k = 0; k < 4; ++k)
  {
    delta[index + (2 * k) * stride] += d_mu[k];
    delta[index + (2 * k + 1) * stride] += d_sg[k];
  }
  return iou;
}

// DeltaGaussianYoloClass
void gaussian_class_delta(const float* output, float* delta, int index, int class_id, int classes, int stride,
    float label_smooth_eps)
{
  if (delta[index])
  {
    float y_true = 1;
    if (label_smooth_eps)
      y_true = y_true * (1 - label_smooth_eps) + 0.5 * label_smooth_eps;
    delta[index + stride * class_id] = y_true - output[index + stride * class_id];
    return;
  }
  for (int n = 0; n < classes; ++n)
  {
    float y_true = ((n == class_id) ? 1 : 0);
    if (label_smooth_eps)
      y_true = y_true * (1 - label_smooth_eps) + 0.5 * label_smooth_eps;
    delta[index + stride * n] = y_true - output[index + stride * n];
  }
}

inline int mask_index(const int* mask, int val, int n)
{
  for (int i = 0; i < n; ++i)
    if (mask[i] == val)
      return i;
  return -1;
}
}  // namespace

extern "C" LIB_API float DkGaussianYoloLossHost(const layer* l, int net_w, int net_h, float* out, const float* truth_all,
    float* delta)
{
  const int stride = l->w * l->h;
  const size_t total = (size_t)l->batch * l->outputs;
  memset(delta, 0, total * sizeof(float));
  for (int b = 0; b < l->batch; ++b)
  {
    const float* tb = truth_all + (size_t)b * l->truths;
    // ---- every predictor: no-object gradient unless it overlaps a truth of a confidently predicted class
    for (int j = 0; j < l->h; ++j)
      for (int i = 0; i < l->w; ++i)
        for (int n = 0; n < l->n; ++n)
        {
          const int loc = n * stride + j * l->w + i;
          const int box_index = gentry(l, b, loc, 0);
          const Box pred = gaussian_box(out, l->biases, l->mask[n], box_index, i, j, l->w, l->h, net_w, net_h, stride,
              l->yolo_point);
          float best_match_iou = 0, best_iou = 0;
          int best_t = 0;
          const int class_index = gentry(l, b, loc, 9);
          const int obj_index = gentry(l, b, loc, 8);
          for (int t = 0; t < l->max_boxes; ++t)
          {
            const Box truth{tb[t * 5 + 0], tb[t * 5 + 1], tb[t * 5 + 2], tb[t * 5 + 3]};
            const int class_id = tb[t * 5 + 4];
            if (class_id >= l->classes)
              continue;
            if (!truth.x)
              break;
            int match = 0;   // CompareGaussianYoloClass with conf_thresh 0.25
            for (int c = 0; c < l->classes; ++c)
              if (out[class_index + stride * c] > 0.25f)
              {
                match = 1;
                break;
              }
            const float iou = Box::Iou(pred, truth);
            if (iou > best_match_iou && match == 1)
              best_match_iou = iou;
            if (iou > best_iou)
            {
              best_iou = iou;
              best_t = t;
            }
          }
          delta[obj_index] = l->cls_normalizer * (0 - out[obj_index]);
          if (best_match_iou > l->ignore_thresh)
            delta[obj_index] = 0;
          if (best_iou > l->truth_thresh)
          {
            delta[obj_index] = l->cls_normalizer * (1 - out[obj_index]);
            const int class_id = tb[best_t * 5 + 4];
            gaussian_class_delta(out, delta, class_index, class_id, l->classes, stride, l->label_smooth_eps);
            const Box truth{tb[best_t * 5 + 0], tb[best_t * 5 + 1], tb[best_t * 5 + 2], tb[best_t * 5 + 3]};
            gaussian_box_delta(truth, out, l->biases, l->mask[n], box_index, i, j, l->w, l->h, net_w, net_h, delta,
                (2 - truth.w * truth.h), stride, l->iou_normalizer * 1.0f, l->iou_loss, l->uc_normalizer, l->yolo_point,
                l->max_delta);
          }
        }
    // ---- every truth: its best anchor (and the anchors above iou_thresh) at the truth's cell
    for (int t = 0; t < l->max_boxes; ++t)
    {
      const Box truth{tb[t * 5 + 0], tb[t * 5 + 1], tb[t * 5 + 2], tb[t * 5 + 3]};
      if (!truth.x)
        break;
      float best_iou = 0;
      int best_n = 0;
      int i = (truth.x * l->w);
      int j = (truth.y * l->h);
      if (l->yolo_point == YOLO_LEFT_TOP)
      {
        i = std::min((float)(l->w - 1), std::max(0.f, ((truth.x - truth.w / 2) * l->w)));
        j = std::min((float)(l->h - 1), std::max(0.f, ((truth.y - truth.h / 2) * l->h)));
      }
      else if (l->yolo_point == YOLO_RIGHT_BOTTOM)
      {
        i = std::min((float)(l->w - 1), std::max(0.f, ((truth.x + truth.w / 2) * l->w)));
        j = std::min((float)(l->h - 1), std::max(0.f, ((truth.y + truth.h / 2) * l->h)));
      }
      Box truth_shift = truth;
      truth_shift.x = truth_shift.y = 0;
      for (int n = 0; n < l->total; ++n)
      {
        Box pred{0, 0, l->biases[2 * n] / net_w, l->biases[2 * n + 1] / net_h};
        const float iou = Box::Iou(pred, truth_shift);
        if (iou > best_iou)
        {
          best_iou = iou;
          best_n = n;
        }
      }
      const int class_id = tb[t * 5 + 4];
      auto assign = [&](int anchor, int mask_n) {
        const int loc = mask_n * stride + j * l->w + i;
        const int box_index = gentry(l, b, loc, 0);
        gaussian_box_delta(truth, out, l->biases, anchor, box_index, i, j, l->w, l->h, net_w, net_h, delta,
            (2 - truth.w * truth.h), stride, l->iou_normalizer * 1.0f, l->iou_loss, l->uc_normalizer, l->yolo_point,
            l->max_delta);
        const int obj_index = gentry(l, b, loc, 8);
        delta[obj_index] = 1.0f * l->cls_normalizer * (1 - out[obj_index]);
        gaussian_class_delta(out, delta, gentry(l, b, loc, 9), class_id, l->classes, stride, l->label_smooth_eps);
      };
      const int mask_n = mask_index(l->mask, best_n, l->n);
      if (mask_n >= 0)
        assign(best_n, mask_n);
      for (int n = 0; n < l->total; ++n)
      {
        const int mn = mask_index(l->mask, n, l->n);
        if (mn >= 0 && n != best_n && l->iou_thresh < 1.0f)
        {
          Box pred{0, 0, l->biases[2 * n] / net_w, l->biases[2 * n + 1] / net_h};
          const float iou = iou_kind(pred, truth_shift, l->iou_thresh_kind);
          if (iou > l->iou_thresh)
            assign(n, mn);
        }
      }
    }
    // ---- AveragesGaussianYoloDeltas over every predictor of the image
    for (int j = 0; j < l->h; ++j)
      for (int i = 0; i < l->w; ++i)
        for (int n = 0; n < l->n; ++n)
        {
          const int loc = n * stride + j * l->w + i;
          const int box_index = gentry(l, b, loc, 0), class_index = gentry(l, b, loc, 9);
          int classes_in_one_box = 0;
          for (int c = 0; c < l->classes; ++c)
            if (delta[class_index + stride * c] > 0)
              classes_in_one_box++;
          if (classes_in_one_box > 0)
            for (int e = 0; e < 8; ++e) delta[box_index + e * stride] /= classes_in_one_box;
        }
  }
  // *(l->cost) = pow(mag_array(l->delta, n), 2): float sum of squares, sqrtf, squared in double
  float sum = 0;
  for (size_t k = 0; k < total; ++k) sum += delta[k] * delta[k];
  const float mag = sqrtf(sum);
  return (float)pow((double)mag, 2.0);
}
