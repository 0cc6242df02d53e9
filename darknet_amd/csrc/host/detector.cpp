// detector.cpp -- headless trainer / evaluator harness around the HIP path.
//
// Reference twins (Ravicmoon/darknet src/): TrainDetector detector.cpp:27-315, ValidateDetector
// :326-562 (the mAP routine), Metadata option_list.cpp:13-97, ReadBoxAnnot data.cpp:78-115,
// ReplaceImage2Label utils.cpp:112-118, SaveWeights(net, dir, base, suffix) parser.cpp.
//
// "Headless": the reference decodes, augments and resizes images with OpenCV in loader threads
// (data.cpp, image_opencv.cpp) -- that subsystem is outside the hot path.  Here an image is a binary
// PPM (P6, 8 bit, RGB) already at the network's resolution; it crosses PCIe as bytes and Mat2Image
// (visualize.cpp:26-55) runs on the device (DkNetworkPredictU8).  No jitter / flip / hue: the
// harness exists to drive the path (train, checkpoint, resume, evaluate) and to state an mAP, not to
// reproduce the reference's augmentation stream.  Labels are the reference's text format
// (`class x y w h` per line, normalised centre boxes); checkpoints are the reference's .weights.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "dk_host.h"
#include "dk_internal.h"

// ---------------------------------------------------------------------------
// Metadata (.data file: classes, train, valid, names, save) -- option_list.cpp:38-69
// ---------------------------------------------------------------------------
static std::vector<std::string> read_lines(const std::string& path)
{
  std::vector<std::string> out;
  FILE* f = fopen(path.c_str(), "r");
  if (!f)
    return out;
  char buf[4096];
  while (fgets(buf, sizeof(buf), f))
  {
    std::string s(buf);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
    if (!s.empty())
      out.push_back(s);
  }
  fclose(f);
  return out;
}

bool Metadata::Get(std::string filename)
{
  std::vector<Section> secs;
  // a .data file is a cfg without section headers: reuse the option reader on a synthetic section
  FILE* f = fopen(filename.c_str(), "r");
  if (!f)
    return false;
  classes_ = 2;
  train_file_ = "train.txt";
  val_file_ = "valid.txt";
  name_file_ = "name.txt";
  save_dir_ = "save";
  char buf[4096];
  while (fgets(buf, sizeof(buf), f))
  {
    std::string s;
    for (char* p = buf; *p; ++p)
      if (*p != ' ' && *p != '\t' && *p != '\n' && *p != '\r')
        s.push_back(*p);
    if (s.empty() || s[0] == '#' || s[0] == ';')
      continue;
    const size_t eq = s.find('=');
    if (eq == std::string::npos)
      continue;
    const std::string k = s.substr(0, eq), v = s.substr(eq + 1);
    if (k == "classes") classes_ = atoi(v.c_str());
    else if (k == "train") train_file_ = v;
    else if (k == "valid") val_file_ = v;
    else if (k == "names") name_file_ = v;
    else if (k == "save") save_dir_ = v;
  }
  fclose(f);
  train_img_list_ = read_lines(train_file_);
  val_img_list_ = read_lines(val_file_);
  name_list_ = read_lines(name_file_);
  return true;
}

// ---------------------------------------------------------------------------
// labels and images
// ---------------------------------------------------------------------------
std::string ReplaceImage2Label(std::string str)
{
  const size_t idx = str.find_last_of('.');
  if (idx != std::string::npos)
    str.replace(str.begin() + idx, str.end(), ".txt");
  return str;
}

std::vector<BoxLabel> ReadBoxAnnot(std::string filename)
{
  std::vector<BoxLabel> annot;
  FILE* file = fopen(filename.c_str(), "r");
  if (!file)
  {
    fprintf(stderr, "Cannot open label file: %s\n", filename.c_str());
    return annot;
  }
  int id = 0;
  float x = 0, y = 0, w = 0, h = 0;
  while (fscanf(file, "%d %f %f %f %f", &id, &x, &y, &w, &h) == 5)
  {
    BoxLabel b;
    b.id = id; b.x = x; b.y = y; b.w = w; b.h = h;
    b.left = x - w / 2; b.right = x + w / 2; b.top = y - h / 2; b.bottom = y + h / 2;
    annot.push_back(b);
  }
  fclose(file);
  return annot;
}

// binary PPM (P6, maxval 255) -> interleaved RGB bytes; returns false on any mismatch
static bool read_ppm(const std::string& path, int want_w, int want_h, unsigned char* dst)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f)
    return false;
  char magic[3] = {0};
  int w = 0, h = 0, maxv = 0;
  auto next_int = [&](int* v) {
    int c = fgetc(f);
    for (;;)
    {
      while (c == ' ' || c == '\n' || c == '\r' || c == '\t') c = fgetc(f);
      if (c == '#')
      {
        while (c != '\n' && c != EOF) c = fgetc(f);
        continue;
      }
      break;
    }
    int x = 0, n = 0;
    while (c >= '0' && c <= '9')
    {
      x = x * 10 + (c - '0');
      c = fgetc(f);
      ++n;
    }
    *v = x;
    return n > 0;   // the single whitespace after the number has been consumed
  };
  bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '6' && next_int(&w) && next_int(&h) && next_int(&maxv);
  ok = ok && w == want_w && h == want_h && maxv == 255;
  if (ok)
    ok = fread(dst, 1, (size_t)w * h * 3, f) == (size_t)w * h * 3;
  fclose(f);
  return ok;
}

// ---------------------------------------------------------------------------
// mAP (detector.cpp:326-562), separated from IO so that it can be checked on its own
// ---------------------------------------------------------------------------
namespace
{
struct ValBox
{
  Box b;
  float p;
  int cid;
  int gt_idx;
  bool matched;
};
}  // namespace

// dets: per image n_dets[i] detections AFTER NmsSort, each [x, y, w, h, prob[classes]];
// gts: per image n_gts[i] labels, each [id, x, y, w, h].  ap_out (may be NULL): classes doubles.
extern "C" LIB_API double DkMeanAveragePrecision(int n_images, const int* n_dets, const float* dets, const int* n_gts,
    const float* gts, int classes, float iou_thresh, double* ap_out)
{
  std::vector<ValBox> val_boxes;
  std::vector<int> num_gt_class(classes, 0), num_pred_class(classes, 0);
  int num_gt = 0;
  const int rec = 4 + classes;
  const float* dp = dets;
  const float* gp = gts;
  for (int i = 0; i < n_images; ++i)
  {
    const int ng = n_gts[i];
    for (int k = 0; k < ng; ++k)
    {
      const int id = (int)gp[5 * k];
      if (id >= 0 && id < classes)
        num_gt_class[id]++;
    }
    for (int j = 0; j < n_dets[i]; ++j)
    {
      const float* d = dp + (size_t)j * rec;
      const Box pred_box(d[0], d[1], d[2], d[3]);
      for (int cid = 0; cid < classes; ++cid)
      {
        const float pred_prob = d[4 + cid];
        if (fabsf(pred_prob) < 1.1920929e-07f)   // FLT_EPSILON
          continue;
        num_pred_class[cid]++;
        int gt_idx = -1;
        float max_iou = 0;
        for (int k = 0; k < ng; ++k)
        {
          const Box gt_box(gp[5 * k + 1], gp[5 * k + 2], gp[5 * k + 3], gp[5 * k + 4]);
          const float iou = Box::Iou(pred_box, gt_box);
          if (iou > iou_thresh && iou > max_iou && cid == (int)gp[5 * k])
          {
            max_iou = iou;
            gt_idx = num_gt + k;
          }
        }
        ValBox v;
        v.b = pred_box; v.p = pred_prob; v.cid = cid; v.matched = gt_idx > -1; v.gt_idx = gt_idx;
        val_boxes.push_back(v);
      }
    }
    num_gt += ng;
    dp += (size_t)n_dets[i] * rec;
    gp += (size_t)ng * 5;
  }
  // precision-recall curve; the reference sorts with std::sort on p alone (ties are ordered
  // arbitrarily there): stable here, by insertion order within equal p
  std::stable_sort(val_boxes.begin(), val_boxes.end(), [](const ValBox& a, const ValBox& b) { return a.p > b.p; });
  const size_t nb = val_boxes.size();
  std::vector<int> tp(classes, 0), fp(classes, 0);
  // the AP integral walks the curve from the end; keep per class only the points where the class's
  // own counters change (the other points repeat the previous precision/recall of that class and
  // contribute delta_recall = 0 with an unchanged running maximum)
  std::vector<std::vector<std::pair<double, double>>> pr(classes);   // (precision, recall) per change
  std::vector<bool> gt_flags(num_gt > 0 ? num_gt : 1, false);
  auto point = [&](int cid) {
    const int t = tp[cid], f = fp[cid], fn = num_gt_class[cid] - t;
    const double precision = (t + f > 0) ? (double)t / (t + f) : 0;
    const double recall = (t + fn > 0) ? (double)t / (t + fn) : 0;
    return std::make_pair(precision, recall);
  };
  for (int cid = 0; cid < classes; ++cid) pr[cid].push_back(point(cid));   // state before any prediction (all zero)
  for (size_t i = 0; i < nb; ++i)
  {
    const ValBox& v = val_boxes[i];
    if (v.matched && !gt_flags[v.gt_idx])
    {
      gt_flags[v.gt_idx] = true;
      tp[v.cid]++;
    }
    else
      fp[v.cid]++;
    pr[v.cid].push_back(point(v.cid));
  }
  double map = 0;
  for (int cid = 0; cid < classes; ++cid)
  {
    double ap = 0;
    if (nb > 0)
    {
      // pr[cid] without its first entry is the reference's pr[cid][0..nb) with repeats removed,
      // EXCEPT that index 0 of the reference's array holds the state after the first prediction
      // overall (zero counters for every class but that prediction's): represented by entry 0 here.
      const std::vector<std::pair<double, double>>& c = pr[cid];
      double last_recall = c.back().second, last_precision = c.back().first;
      const size_t first = (val_boxes[0].cid == cid) ? 1 : 0;   // entry 0 exists in the reference's curve only for the other classes
      for (size_t k = c.size(); k-- > first;)
      {
        const double delta_recall = last_recall - c[k].second;
        last_recall = c[k].second;
        last_precision = last_precision > c[k].first ? last_precision : c[k].first;
        ap += delta_recall * last_precision;
      }
    }
    if (ap_out)
      ap_out[cid] = ap;
    map += ap;
  }
  return classes > 0 ? map / classes : 0;
}

// ---------------------------------------------------------------------------
// ValidateDetector
// ---------------------------------------------------------------------------
static double now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

float DkValidateDetector(Metadata const& md, Network* net, float iou_thresh, float thresh, float nms)
{
  std::vector<std::string> names = md.NameList();
  layer* l = &net->layers[net->n - 1];
  const int classes = l->classes;
  if (classes != (int)names.size())
  {
    printf(" # of names %d is not equal to  # of classes %d\n", (int)names.size(), classes);
    return -1.0f;
  }
  if (net->c != 3)
    error("ValidateDetector: PPM input needs a 3-channel network");
  std::vector<std::string> imgs = md.ValImgList();
  const int B = net->batch;
  const size_t frame = (size_t)net->w * net->h * 3;
  std::vector<unsigned char> frames(frame * B);
  std::vector<int> n_dets, n_gts;
  std::vector<float> dets, gts;
  const int rec = 4 + classes;
  double pred_time = 0;
  for (size_t i0 = 0; i0 < imgs.size(); i0 += B)
  {
    const int nb = (int)std::min<size_t>(B, imgs.size() - i0);
    for (int b = 0; b < B; ++b)
    {
      const std::string& path = imgs[i0 + (b < nb ? b : 0)];   // a short last batch repeats its first image
      if (!read_ppm(path, net->w, net->h, frames.data() + frame * b))
      {
        fprintf(stderr, "ValidateDetector: %s is not a %dx%d binary PPM (images must be at the network's resolution)\n",
            path.c_str(), net->w, net->h);
        exit(EXIT_FAILURE);
      }
    }
    const double t0 = now_s();
    DkNetworkPredictU8(net, frames.data(), (size_t)net->w * 3);
    NetworkSync(net);
    pred_time += now_s() - t0;
    for (int b = 0; b < nb; ++b)
    {
      int num = 0;
      Detection* d = GetNetworkBoxesBatch(net, b, thresh, &num);
      NmsSort(d, num, classes, nms, l->nms_kind, l->beta_nms);
      n_dets.push_back(num);
      for (int j = 0; j < num; ++j)
      {
        dets.push_back(d[j].bbox.x); dets.push_back(d[j].bbox.y); dets.push_back(d[j].bbox.w); dets.push_back(d[j].bbox.h);
        dets.insert(dets.end(), d[j].prob, d[j].prob + classes);
      }
      FreeDetections(d, num);
      std::vector<BoxLabel> gt = ReadBoxAnnot(ReplaceImage2Label(imgs[i0 + b]));
      n_gts.push_back((int)gt.size());
      for (auto& g : gt)
      {
        gts.push_back((float)g.id); gts.push_back(g.x); gts.push_back(g.y); gts.push_back(g.w); gts.push_back(g.h);
      }
    }
  }
  (void)rec;
  std::vector<double> ap(classes, 0);
  const double map = DkMeanAveragePrecision((int)n_dets.size(), n_dets.data(), dets.data(), n_gts.data(), gts.data(),
      classes, iou_thresh, ap.data());
  for (int cid = 0; cid < classes; ++cid) printf(" cid = %d, name = %s, ap = %.4f%%\n", cid, names[cid].c_str(), ap[cid] * 100);
  printf("\n mAP@%g: %.4f%%\n\n Total prediction time: %gs\n Prediction per second: %g\n", iou_thresh, map * 100,
      pred_time, pred_time > 0 ? imgs.size() / pred_time : 0.0);
  return (float)map;
}

float ValidateDetector(Metadata const& md, Network* net, float const iou_thresh)
{
  return DkValidateDetector(md, net, iou_thresh, .005f, .45f);   // detector.cpp:340-341
}

// ---------------------------------------------------------------------------
// TrainDetector
// ---------------------------------------------------------------------------
static void save_named(Network* net, const std::string& dir, const std::string& base, const std::string& suffix)
{
  const std::string path = dir + "/" + base + "_" + suffix + ".weights";
  SaveWeights(net, path.c_str());
  printf("Saving weights to %s\n", path.c_str());
}

int GetCurrIter(Network* net) { return net->curr_iter; }

void DkTrainDetector(Metadata const& md, std::string model_file, std::string weights_file, int num_gpus, bool clear,
    bool calc_map, int max_iterations, int save_every, float map_thresh)
{
  const std::string save_dir = md.SaveDir();
  mkdir(save_dir.c_str(), 0755);
  if (num_gpus < 1)
    num_gpus = 1;
  if (CudaGetDeviceCount() < num_gpus)
    error("TrainDetector: fewer HIP devices than num_gpus");
  Network* nets = (Network*)xcalloc(num_gpus, sizeof(Network));
  for (int k = 0; k < num_gpus; ++k)
  {
    cuda_set_device(k);
    if (!LoadNetwork(nets + k, model_file.c_str(), weights_file.empty() ? nullptr : weights_file.c_str(), true, clear))
      error("TrainDetector: cannot load the network");
    // the reference multiplies lr by num_gpus for its asynchronous replicas (detector.cpp:67); the
    // synchronous all-reduce here normalises by the global batch instead, lr stays as configured
  }
  Network* net = &nets[0];
  printf("Learning rate: %e, Momentum: %g, Decay: %g\n", net->lr, net->momentum, net->decay);
  std::vector<std::string> paths = md.TrainImgList();
  const int num_train = (int)paths.size();
  if (!num_train)
    error("TrainDetector: empty training list");
  const int actual_batch = net->batch * net->subdiv;
  const int img_per_step = actual_batch * num_gpus;
  const int iter_per_epoch = (int)((float)num_train / actual_batch + 0.5f);
  for (int k = 0; k < num_gpus; ++k) nets[k].max_iter = iter_per_epoch * net->max_epoch;
  if (max_iterations > 0)
    for (int k = 0; k < num_gpus; ++k) nets[k].max_iter = std::min(nets[k].max_iter > 0 ? nets[k].max_iter : max_iterations, max_iterations);
  printf("Max number of iterations: %d\n", net->max_iter);
  layer* l = &net->layers[net->n - 1];
  const int truths = net->truths > 0 ? net->truths : l->max_boxes * 5;
  const size_t frame = (size_t)net->w * net->h * 3;
  std::vector<unsigned char> u8(frame);
  std::vector<float> X((size_t)img_per_step * net->inputs), Y((size_t)img_per_step * truths);
  std::vector<float*> xv(img_per_step), yv(img_per_step);
  for (int r = 0; r < img_per_step; ++r)
  {
    xv[r] = X.data() + (size_t)r * net->inputs;
    yv[r] = Y.data() + (size_t)r * truths;
  }
  const size_t last_dot = model_file.find_last_of('.');
  const size_t last_slash = model_file.find_last_of('/');
  const std::string base = model_file.substr(last_slash == std::string::npos ? 0 : last_slash + 1,
      (last_dot == std::string::npos ? model_file.size() : last_dot) - (last_slash == std::string::npos ? 0 : last_slash + 1));
  int iter_save = GetCurrIter(net);
  float avg_loss = -1, best_map = 0;
  size_t cursor = ((size_t)GetCurrIter(net) * img_per_step) % num_train;   // resume where the list left off
  if (save_every < 1)
    save_every = 1000;
  while (GetCurrIter(net) < net->max_iter)
  {
    // next img_per_step images in list order (the reference draws them at random with augmentation)
    std::fill(Y.begin(), Y.end(), 0.f);
    for (int r = 0; r < img_per_step; ++r)
    {
      const std::string& path = paths[cursor];
      cursor = (cursor + 1) % num_train;
      if (!read_ppm(path, net->w, net->h, u8.data()))
      {
        fprintf(stderr, "TrainDetector: %s is not a %dx%d binary PPM\n", path.c_str(), net->w, net->h);
        exit(EXIT_FAILURE);
      }
      float* x = xv[r];
      const size_t hw = (size_t)net->w * net->h;
      for (size_t p = 0; p < hw; ++p)   // Mat2Image, visualize.cpp:26-55
        for (int k = 0; k < 3; ++k) x[k * hw + p] = u8[p * 3 + k] / 255.0f;
      std::vector<BoxLabel> gt = ReadBoxAnnot(ReplaceImage2Label(path));
      float* y = yv[r];
      int n = 0;
      for (auto& g : gt)
      {
        if (n >= l->max_boxes || g.id >= l->classes || g.id < 0)
          continue;
        y[n * 5 + 0] = g.x; y[n * 5 + 1] = g.y; y[n * 5 + 2] = g.w; y[n * 5 + 3] = g.h; y[n * 5 + 4] = (float)g.id;
        ++n;
      }
    }
    data d;
    memset(&d, 0, sizeof(d));
    d.X.rows = d.y.rows = img_per_step;
    d.X.cols = net->inputs;
    d.y.cols = truths;
    d.X.vals = xv.data();
    d.y.vals = yv.data();
    d.shallow = 1;
    const double t0 = now_s();
    const float loss = num_gpus == 1 ? TrainNetwork(net, d) : TrainNetworks(nets, num_gpus, d, 4);
    if (avg_loss < 0)
      avg_loss = loss;
    avg_loss = avg_loss * 0.9f + loss * 0.1f;
    const int iter = GetCurrIter(net);
    printf("[%04d] loss: %.2f, avg loss: %.2f, lr: %e, images: %d, %.3f s/iter\n", iter, loss, avg_loss, GetCurrLr(net),
        iter * img_per_step, now_s() - t0);
    if (iter >= iter_save + save_every || iter % save_every == 0)
    {
      iter_save = iter;
      if (num_gpus != 1)
        SyncNetworks(nets, num_gpus);
      save_named(net, save_dir, base, std::to_string(iter));
    }
  }
  if (num_gpus != 1)
    SyncNetworks(nets, num_gpus);
  save_named(net, save_dir, base, "final");
  if (calc_map)
  {
    // evaluate the trained weights with an inference load of the same cfg (BN folded)
    const std::string fin = save_dir + "/" + base + "_final.weights";
    cuda_set_device(0);
    Network* nm = (Network*)xcalloc(1, sizeof(Network));
    if (!LoadNetworkBatch(nm, model_file.c_str(), fin.c_str(), 1))
      error("TrainDetector: cannot reload the final weights");
    const float map = DkValidateDetector(md, nm, 0.5f, map_thresh > 0 ? map_thresh : .005f, .45f);
    if (map > best_map)
      best_map = map;
    printf("Best mAP = %g\n", best_map);
    FreeNetwork(nm);
    free(nm);
  }
  for (int k = 0; k < num_gpus; ++k) FreeNetwork(&nets[k]);
  free(nets);
}

void TrainDetector(Metadata const& md, std::string model_file, std::string weights_file, int num_gpus, bool clear,
    bool show_imgs, bool calc_map, int benchmark_layers)
{
  (void)show_imgs;
  (void)benchmark_layers;
  DkTrainDetector(md, model_file, weights_file, num_gpus, clear, calc_map, 0, 1000, 0.f);
}

// ---- flat helpers for FFI callers (ctypes) -------------------------------------------
extern "C" LIB_API float DkValidateDetectorFlat(const char* data_file, Network* net, float iou_thresh, float thresh, float nms)
{
  Metadata md;
  if (!md.Get(data_file))
    error("DkValidateDetectorFlat: cannot read the .data file");
  return DkValidateDetector(md, net, iou_thresh, thresh, nms);
}

extern "C" LIB_API void DkTrainDetectorFlat(const char* data_file, const char* cfg, const char* weights, int num_gpus, int clear,
    int calc_map, int max_iterations, int save_every, float map_thresh)
{
  Metadata md;
  if (!md.Get(data_file))
    error("DkTrainDetectorFlat: cannot read the .data file");
  DkTrainDetector(md, cfg, weights ? weights : "", num_gpus, clear != 0, calc_map != 0, max_iterations, save_every, map_thresh);
}
