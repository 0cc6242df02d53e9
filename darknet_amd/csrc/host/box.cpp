// box.cpp -- post-processing kept as host C++ (SURVEY.md section 8f-1 marks a
// device version as "next"): IoU family needed by NMS, NmsSort, GetMostProbDets.
// Reference: src/box.cpp:36-71 (Overlap/Intersect/Union/Iou), :98-113 (Diou),
// :372-419 (NmsComparator, NmsSort), :421-447 (GetMostProbDets).
#include <float.h>
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "dk_host.h"

float Box::Overlap(float x1, float w1, float x2, float w2)
{
  const float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
  const float left = l1 > l2 ? l1 : l2;
  const float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
  const float right = r1 < r2 ? r1 : r2;
  return right - left;
}

float Box::Intersect(Box const& b1, Box const& b2)
{
  const float w = Overlap(b1.x, b1.w, b2.x, b2.w);
  const float h = Overlap(b1.y, b1.h, b2.y, b2.h);
  if (w < 0 || h < 0)
    return 0;
  return w * h;
}

float Box::Union(Box const& b1, Box const& b2)
{
  return b1.w * b1.h + b2.w * b2.h - Intersect(b1, b2);
}

float Box::Iou(Box const& b1, Box const& b2)
{
  const float I = Intersect(b1, b2);
  const float U = Union(b1, b2);
  if (fabsf(I) < FLT_EPSILON || fabsf(U) < FLT_EPSILON)
    return 0;
  return I / U;
}

float Box::Diou(Box const& b1, Box const& b2, float beta)
{
  // minimum enclosing box (Box::AbsBox, src/box.cpp:352-370)
  const float left = fminf(b1.x - b1.w / 2.0f, b2.x - b2.w / 2.0f);
  const float right = fmaxf(b1.x + b1.w / 2.0f, b2.x + b2.w / 2.0f);
  const float top = fminf(b1.y - b1.h / 2.0f, b2.y - b2.h / 2.0f);
  const float bottom = fmaxf(b1.y + b1.h / 2.0f, b2.y + b2.h / 2.0f);
  const float w = right - left, h = bottom - top;
  const float c = w * w + h * h;
  const float iou = Iou(b1, b2);
  if (fabsf(c) < FLT_EPSILON)
    return iou;
  const float d = (b1.x - b2.x) * (b1.x - b2.x) + (b1.y - b2.y) * (b1.y - b2.y);
  const float diou_term = powf(d / c, beta);
  return iou - diou_term;
}

// NmsSort: per class, sort by that class' probability (descending) and zero the
// probability of every later box whose IoU (or DIoU) with an earlier, still
// non-zero box exceeds thresh.  The reference sorts with qsort (glibc: a stable
// merge sort); std::stable_sort reproduces the order for ties.
void NmsSort(Detection* dets, int total, int classes, float thresh, NMS_KIND nms_kind, float beta)
{
  for (int k = 0; k < classes; ++k)
  {
    for (int i = 0; i < total; ++i) dets[i].sort_class = k;
    std::stable_sort(dets, dets + total,
        [k](const Detection& a, const Detection& b) { return a.prob[k] > b.prob[k]; });
    for (int i = 0; i < total; ++i)
    {
      if (fabsf(dets[i].prob[k]) < FLT_EPSILON)
        continue;
      const Box a = dets[i].bbox;
      for (int j = i + 1; j < total; ++j)
      {
        const Box b = dets[j].bbox;
        if (nms_kind == GREEDY_NMS && Box::Iou(a, b) > thresh)
          dets[j].prob[k] = 0.0f;
        else if (nms_kind == DIOU_NMS && Box::Diou(a, b, beta) > thresh)
          dets[j].prob[k] = 0.0f;
      }
    }
  }
}

std::vector<MostProbDet> GetMostProbDets(Detection* dets, int num_dets)
{
  std::vector<MostProbDet> out;
  for (int i = 0; i < num_dets; i++)
  {
    int cid = -1;
    float max_prob = 0.0f;
    for (int j = 0; j < dets[i].classes; j++)
      if (dets[i].prob[j] > max_prob)
      {
        cid = j;
        max_prob = dets[i].prob[j];
      }
    if (cid != -1)
    {
      MostProbDet m;
      m.bbox = dets[i].bbox;
      m.cid = cid;
      m.prob = max_prob;
      out.push_back(m);
    }
  }
  return out;
}

// Flat NMS for FFI callers: buf holds `num` records [x,y,w,h,obj,prob[classes]];
// sorted/suppressed in place exactly like NmsSort on Detection structs.
extern "C" LIB_API void DkNmsSortFlat(
    float* buf, int num, int classes, float thresh, int nms_kind, float beta)
{
  const int rec = 5 + classes;
  std::vector<Detection> dets(num);
  std::vector<float> probs((size_t)num * classes);
  for (int i = 0; i < num; ++i)
  {
    float* o = buf + (size_t)i * rec;
    dets[i].bbox = Box(o[0], o[1], o[2], o[3]);
    dets[i].objectness = o[4];
    dets[i].classes = classes;
    dets[i].prob = probs.data() + (size_t)i * classes;
    for (int j = 0; j < classes; ++j) dets[i].prob[j] = o[5 + j];
  }
  NmsSort(dets.data(), num, classes, thresh, (NMS_KIND)nms_kind, beta);
  std::vector<float> tmp((size_t)num * rec);
  for (int i = 0; i < num; ++i)
  {
    float* o = tmp.data() + (size_t)i * rec;
    o[0] = dets[i].bbox.x; o[1] = dets[i].bbox.y; o[2] = dets[i].bbox.w; o[3] = dets[i].bbox.h;
    o[4] = dets[i].objectness;
    for (int j = 0; j < classes; ++j) o[5 + j] = dets[i].prob[j];
  }
  std::copy(tmp.begin(), tmp.end(), buf);
}
