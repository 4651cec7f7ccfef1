// Native host helper of the optimizer shells: the per-tensor pointer-aliasing loops in C++ (the layers' autograd
// nodes live in host_autograd.cpp, same module).
//
// The reference re-points `param.data` (and hands over `param.grad`) tensor by tensor in Python
// (src/algos/svgd.py:93-96,120-127; swag.py:58,81; ivorn.py:111).  With ResNet-50's 161 tensors an
// SVGD step does that 2 * 8 * 161 times -- ~4 ms of interpreter time around 0.5 ms of kernels.  Here
// the same loop runs over the already-unpacked tensor lists.  No arithmetic, no device work: plumbing.
#include <torch/extension.h>

#include <vector>

// the C ABI the native calls below go through: their function-pointer types are TAKEN from these declarations
// (decltype(&bde_...)), so an edit of the header that this file does not follow fails to compile instead of passing
// arguments in the wrong slots; tests/test_abi.py pins the same three signatures against _lib.py
#include "../../include/bde_hip.h"

namespace {

// param.data = view.  All particle views of one parameter live in ONE storage with ONE shape (rows of the flat
// particle buffer), so switching particle is a change of the parameter's storage offset: one store instead of
// set_data()'s shallow copy of the whole TensorImpl (0.45 us per tensor on the GPU box: 8 x 161 of them were 0.58 ms
// of a ResNet-50 SVGD step).  Anything else (first use, a parameter the caller re-pointed elsewhere) takes set_data().
inline void point_at(at::Tensor& p, const at::Tensor& view) {
  c10::TensorImpl* pi = p.unsafeGetTensorImpl();
  const c10::TensorImpl* vi = view.unsafeGetTensorImpl();
  if (pi->storage().unsafeGetStorageImpl() == vi->storage().unsafeGetStorageImpl() && pi->allow_tensor_metadata_change() &&
      pi->dtype() == vi->dtype() && pi->sizes() == vi->sizes() && pi->strides() == vi->strides())
    pi->set_storage_offset(vi->storage_offset());
  else
    p.set_data(view);
}

// param[i].data = datas[i] (if given); param[i].grad = grads[i] (if given; None clears).
void repoint(const std::vector<at::Tensor>& params, const c10::optional<std::vector<at::Tensor>>& datas,
             const c10::optional<std::vector<at::Tensor>>& grads) {
  const size_t n = params.size();
  if (datas.has_value()) {
    TORCH_CHECK(datas->size() == n, "repoint: ", n, " parameters but ", datas->size(), " data tensors");
    for (size_t i = 0; i < n; ++i) {
      at::Tensor p = params[i];
      p.set_data((*datas)[i]);
    }
  }
  if (grads.has_value()) {
    TORCH_CHECK(grads->size() == n, "repoint: ", n, " parameters but ", grads->size(), " gradient tensors");
    for (size_t i = 0; i < n; ++i) {
      at::Tensor p = params[i];
      p.mutable_grad() = (*grads)[i];
    }
  }
}

// After a backward pass: move the gradients autograd produced into `views` (rows of the flat gradient buffer)
// with ONE multi-tensor copy (or add), zero the views of parameters that received none, and make the views the
// parameters' .grad.  Same semantics as algo.adopt_grads.
void adopt_grads(const std::vector<at::Tensor>& params, const std::vector<at::Tensor>& views, bool add) {
  const size_t n = params.size();
  TORCH_CHECK(views.size() == n, "adopt_grads: ", n, " parameters but ", views.size(), " views");
  std::vector<at::Tensor> src, dst, missing;
  src.reserve(n);
  dst.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    const at::Tensor& g = params[i].grad();
    if (!g.defined()) {
      if (!add) missing.push_back(views[i]);
    } else if (g.data_ptr() != views[i].data_ptr()) {
      src.push_back(g);
      dst.push_back(views[i]);
    }
  }
  {
    at::NoGradGuard no_grad;
    if (!missing.empty()) at::_foreach_zero_(missing);
    if (!src.empty()) {
      if (add) at::_foreach_add_(dst, src);
      else at::_foreach_copy_(dst, src);
    }
  }
  for (size_t i = 0; i < n; ++i) {
    at::Tensor p = params[i];
    p.mutable_grad() = views[i];
  }
}

// _store_grads (svgd.py:129-133) WITHOUT the clones: after a backward pass, record WHERE each parameter's gradient
// lives instead of copying it.  table[i * M + j] (a host int64 tensor, pinned when on a GPU box; i = parameter,
// j = particle) receives the address of the gradient tensor autograd produced when the update kernels can read it in
// place (fp32, contiguous, 16-byte aligned, on the views' device); otherwise the gradient is copied into (a missing one
// zeroes) its view of the flat gradient row and the view's address is recorded.  With `zero_addr` != 0 (the address
// of a read-only all-zero buffer at least as long as the largest tensor) a MISSING gradient costs nothing: that
// address is recorded and nothing is written.  Returns the tensors recorded by reference: the caller keeps them alive
// until the update kernel has been enqueued.
// `take`: the gradients are also detached from the parameters (param.grad = None afterwards; the ones recorded by
// reference are MOVED out of the parameter, no reference-count traffic).
std::vector<at::Tensor> collect_grads(const std::vector<at::Tensor>& params, const std::vector<at::Tensor>& views,
                                      at::Tensor table, int64_t j, int64_t M, int64_t zero_addr, bool take) {
  const size_t n = params.size();
  TORCH_CHECK(views.size() == n, "collect_grads: ", n, " parameters but ", views.size(), " views");
  TORCH_CHECK(table.scalar_type() == at::kLong && table.is_contiguous() && !table.is_cuda() &&
                  table.numel() >= static_cast<int64_t>(n) * M && j >= 0 && j < M,
              "collect_grads: table must be a contiguous host int64 tensor with n_tensors * M entries");
  int64_t* t = table.data_ptr<int64_t>();
  std::vector<at::Tensor> keep, src, dst, missing;
  keep.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    at::Tensor& g = params[i].mutable_grad();      // the slot itself (Tensor::grad() adds a leaf check per call)
    const at::Tensor& v = views[i];
    int64_t addr = reinterpret_cast<int64_t>(v.data_ptr());
    if (!g.defined()) {
      if (zero_addr != 0) addr = zero_addr;
      else missing.push_back(v);
    } else if (g.data_ptr() == v.data_ptr()) {
      if (take) g = at::Tensor();                  // accumulated in place: already where it belongs
    } else if (g.scalar_type() == at::kFloat && g.layout() == at::kStrided && g.is_contiguous() && g.device() == v.device() &&
               g.numel() == v.numel() && (reinterpret_cast<uintptr_t>(g.data_ptr()) & 15u) == 0) {
      addr = reinterpret_cast<int64_t>(g.data_ptr());
      if (take) keep.push_back(std::move(g));      // leaves param.grad undefined
      else keep.push_back(g);
    } else {
      src.push_back(g);
      dst.push_back(v);
      if (take) g = at::Tensor();
    }
    t[static_cast<int64_t>(i) * M + j] = addr;
  }
  at::NoGradGuard no_grad;
  if (!missing.empty()) at::_foreach_zero_(missing);
  if (!src.empty()) at::_foreach_copy_(dst, src);
  return keep;
}

void clear_grads(const std::vector<at::Tensor>& params) {
  for (const at::Tensor& t : params) {
    at::Tensor p = t;
    p.mutable_grad() = at::Tensor();
  }
}

// Offsets / shapes of a list of parameters inside one flat row (algo.FlatLayout): per-parameter views of a row, and
// "point the parameters at this row" (vector_to_parameters without the copy, swag.py:58), as single calls.
class Layout {
 public:
  Layout(std::vector<int64_t> offsets, std::vector<int64_t> numels, std::vector<std::vector<int64_t>> shapes)
      : offsets_(std::move(offsets)), numels_(std::move(numels)), shapes_(std::move(shapes)) {
    TORCH_CHECK(offsets_.size() == numels_.size() && numels_.size() == shapes_.size(), "Layout: ragged description");
  }

  std::vector<at::Tensor> views(const at::Tensor& row) const {
    std::vector<at::Tensor> out;
    out.reserve(offsets_.size());
    for (size_t i = 0; i < offsets_.size(); ++i) out.push_back(row.narrow(0, offsets_[i], numels_[i]).view(shapes_[i]));
    return out;
  }

  void point_data(const std::vector<at::Tensor>& params, const at::Tensor& row) const {
    TORCH_CHECK(params.size() == offsets_.size(), "Layout.point_data: ", offsets_.size(), " tensors in the layout but ",
                params.size(), " parameters");
    const c10::StorageImpl* storage = row.unsafeGetTensorImpl()->storage().unsafeGetStorageImpl();
    const int64_t row0 = row.storage_offset();
    const bool row_ok = row.dim() == 1 && row.stride(0) == 1;
    for (size_t i = 0; i < offsets_.size(); ++i) {
      at::Tensor p = params[i];
      c10::TensorImpl* pi = p.unsafeGetTensorImpl();
      // already a contiguous view of this shape in the row's storage (another row of the same buffer): move the offset
      if (row_ok && pi->storage().unsafeGetStorageImpl() == storage && pi->allow_tensor_metadata_change() &&
          pi->dtype() == row.dtype() && pi->is_contiguous() && pi->sizes() == c10::IntArrayRef(shapes_[i]) &&
          offsets_[i] + numels_[i] <= row.numel())
        pi->set_storage_offset(row0 + offsets_[i]);
      else
        p.set_data(row.narrow(0, offsets_[i], numels_[i]).view(shapes_[i]));
    }
  }

 private:
  std::vector<int64_t> offsets_, numels_;
  std::vector<std::vector<int64_t>> shapes_;
};

// The per-particle loops of SVGDOptimizer.step as ONE object that owns its tensor lists (the Python lists are converted
// once, not on every call -- at ResNet-50's 161 tensors the conversions were a third of the step's host time):
//   begin(i)  _use_particle(i) (svgd.py:120-127) + zero_grad(set_to_none) (svgd.py:70): param.data = particle i's view
//   end(i)    _store_grads (svgd.py:129-133) without the clones: collect_grads semantics; the gradients are then
//             detached from the parameters (param.grad = None), so the next backward hands over fresh tensors again
//   release() the gradient tensors taken by reference go back to the allocator (after the update was enqueued)
//   set_grads(i) / use(i): the hand-over of -phi rows to the base optimizer (svgd.py:92-96)
class ParticleSet {
 public:
  ParticleSet(std::vector<at::Tensor> params, std::vector<std::vector<at::Tensor>> pviews,
              std::vector<std::vector<at::Tensor>> gviews)
      : params_(std::move(params)), pviews_(std::move(pviews)), gviews_(std::move(gviews)) {
    for (const auto& v : pviews_) TORCH_CHECK(v.empty() || v.size() == params_.size(), "ParticleSet: ragged particle views");
    for (const auto& v : gviews_) TORCH_CHECK(v.empty() || v.size() == params_.size(), "ParticleSet: ragged gradient views");
    keep_.resize(pviews_.size());
  }

  void use(int64_t i) {
    const auto& views = views_of(pviews_, i, "particle");
    for (size_t k = 0; k < params_.size(); ++k) point_at(params_[k], views[k]);
  }

  void begin(int64_t i) {
    const auto& views = views_of(pviews_, i, "particle");
    for (size_t k = 0; k < params_.size(); ++k) {
      at::Tensor& p = params_[k];
      point_at(p, views[k]);
      at::Tensor& g = p.mutable_grad();
      if (g.defined()) g = at::Tensor();
    }
  }

  void set_grads(int64_t i) {
    const auto& pv = views_of(pviews_, i, "particle");
    const auto& gv = views_of(gviews_, i, "gradient");
    for (size_t k = 0; k < params_.size(); ++k) {
      point_at(params_[k], pv[k]);
      params_[k].mutable_grad() = gv[k];
    }
  }

  int64_t end(int64_t i, at::Tensor table, int64_t row, int64_t M, int64_t zero_addr) {
    const auto& views = views_of(gviews_, i, "gradient");
    TORCH_CHECK(row >= 0 && row < static_cast<int64_t>(keep_.size()), "ParticleSet.end: row out of range");
    keep_[row] = collect_grads(params_, views, std::move(table), row, M, zero_addr, /*take=*/true);
    return static_cast<int64_t>(keep_[row].size());
  }

  // end(i) immediately followed by begin(next): one call from the interpreter instead of two (the step of a small model
  // is host-bound: 8 particles x 2 calls were a sixth of it)
  int64_t end_begin(int64_t i, at::Tensor table, int64_t row, int64_t M, int64_t zero_addr, int64_t next) {
    const int64_t kept = end(i, std::move(table), row, M, zero_addr);
    begin(next);
    return kept;
  }

  // (Handing the tensors to a worker thread instead was measured: the 8 x 161 frees, 0.28 ms, leave this thread, but the
  // closures get slower by as much -- allocator lock and, for gradients with Python wrappers, the interpreter lock --
  // and the step takes the same time: profiles/r03_shell_host_profile_release_ab.txt.)
  void release() {
    for (auto& k : keep_) k.clear();
  }

 private:
  static const std::vector<at::Tensor>& views_of(const std::vector<std::vector<at::Tensor>>& lists, int64_t i, const char* what) {
    TORCH_CHECK(i >= 0 && i < static_cast<int64_t>(lists.size()) && !lists[i].empty(), "ParticleSet: no ", what,
                " views for particle ", i);
    return lists[i];
  }
  std::vector<at::Tensor> params_;
  std::vector<std::vector<at::Tensor>> pviews_, gviews_;
  std::vector<std::vector<at::Tensor>> keep_;
};

// The value SVGDOptimizer.step returns (svgd.py:66,72,105: the particles' losses summed in order, divided by their count) by
// ONE call of bde_mean_scalars: the per-tensor checks and the pointer array without a Python loop (8 losses: ~9 us of a
// ~65 us step in Python).  `entry` is the address of bde_mean_scalars in the kernel library the caller uses (the device
// library; the tests' CPU model or stub), `stream` its stream handle.  Returns false -- the caller then takes torch's adds
// -- when the losses are not fp32 one-element tensors on `out`'s device (a half-precision or off-device loss).
bool mean_losses(const std::vector<at::Tensor>& losses, at::Tensor out, double divisor, int64_t entry, int64_t stream) {
  const int64_t n = static_cast<int64_t>(losses.size());
  if (n < 1 || n > 64 || entry == 0 || !(divisor > 0.0)) return false;
  if (!out.defined() || out.scalar_type() != at::kFloat || out.numel() != 1) return false;
  const float* ptrs[64];
  for (int64_t i = 0; i < n; ++i) {
    const at::Tensor& t = losses[i];
    if (!t.defined() || t.scalar_type() != at::kFloat || t.numel() != 1 || t.device() != out.device()) return false;
    ptrs[i] = t.data_ptr<float>();
  }
  using Fn = decltype(&bde_mean_scalars);
  const int rc = reinterpret_cast<Fn>(entry)(ptrs, static_cast<int>(n), static_cast<float>(divisor), out.data_ptr<float>(),
                                            reinterpret_cast<void*>(stream));
  TORCH_CHECK(rc == 0, "bde_mean_scalars failed with code ", rc);
  return true;
}

// A small model's posterior update on one GPU is host-bound (~15 us of kernels): its two C-ABI calls -- bde_svgd_gather_seg
// (gradients -> flat rows) and bde_svgd_step_small_sgd / _adam (statistics, -phi, the M shared-state optimizer applications)
// -- issued from ONE native function instead of two Python wrappers + ctypes marshalling (~13 us of a ~55 us step).  The
// entry addresses are those of the kernel library the caller uses (device library, the tests' CPU model, the stub); the
// tensors are the optimizer's own persistent buffers, validated where they are created (ops.py), re-checked cheaply here.
struct SmallStepArgs {
  float *P, *G, *s0, *s1;
  void* ws;
  float* kstat;
  int m;
  int64_t ld;
};

SmallStepArgs small_step_common(int64_t e_gather, const at::Tensor& seg_ptrs, const at::Tensor& chunks, const at::Tensor& P,
                                const at::Tensor& G, const at::Tensor& s0, const c10::optional<at::Tensor>& s1,
                                const at::Tensor& ws, const at::Tensor& kstat, int64_t d, int64_t stream) {
  TORCH_CHECK(P.dim() == 2 && G.dim() == 2 && P.scalar_type() == at::kFloat && G.scalar_type() == at::kFloat &&
                  P.stride(1) == 1 && G.stride(1) == 1 && P.stride(0) == G.stride(0) && P.size(0) == G.size(0) &&
                  d >= 1 && d <= P.size(1), "small_step: P and G must be fp32 [M, ld] with one leading dimension");
  TORCH_CHECK(seg_ptrs.scalar_type() == at::kLong && chunks.scalar_type() == at::kLong && chunks.dim() == 2 &&
                  chunks.size(1) == 4 && chunks.is_contiguous() && seg_ptrs.is_contiguous(), "small_step: segment table");
  const auto dev = P.device();
  TORCH_CHECK(G.device() == dev && s0.device() == dev && ws.device() == dev && kstat.device() == dev &&
                  seg_ptrs.device() == dev && chunks.device() == dev && (!s1 || s1->device() == dev),
              "small_step: every buffer must live on the particles' device");
  TORCH_CHECK(s0.scalar_type() == at::kFloat && s0.numel() >= d && (!s1 || (s1->scalar_type() == at::kFloat && s1->numel() >= d)) &&
                  kstat.scalar_type() == at::kFloat, "small_step: optimizer state / statistics buffers");
  SmallStepArgs a{P.data_ptr<float>(), G.data_ptr<float>(), s0.data_ptr<float>(), s1 ? s1->data_ptr<float>() : nullptr,
                  ws.data_ptr(), kstat.data_ptr<float>(), static_cast<int>(P.size(0)), P.stride(0)};
  using Gather = decltype(&bde_svgd_gather_seg);
  static_assert(sizeof(bde_seg_chunk) == 4 * sizeof(int64_t), "a segment-table row is four int64 (ops.SegTable)");
  const int rc = reinterpret_cast<Gather>(e_gather)(reinterpret_cast<const void* const*>(seg_ptrs.data_ptr<int64_t>()),
                                                    reinterpret_cast<const bde_seg_chunk*>(chunks.data_ptr<int64_t>()),
                                                    chunks.size(0), a.G, a.m, 0, a.m, a.ld,
                                                    reinterpret_cast<void*>(stream));
  TORCH_CHECK(rc == 0, "bde_svgd_gather_seg failed with code ", rc);
  return a;
}

void small_step_sgd(int64_t e_gather, int64_t e_step, at::Tensor seg_ptrs, at::Tensor chunks, at::Tensor P, at::Tensor G,
                    at::Tensor buf, at::Tensor ws, at::Tensor kstat, int64_t d, double l2_reg, double kernel_grad_scale,
                    double dataset_size, double lr, double momentum, double dampening, double weight_decay, bool nesterov,
                    bool first, int64_t stream) {
  const SmallStepArgs a = small_step_common(e_gather, seg_ptrs, chunks, P, G, buf, c10::nullopt, ws, kstat, d, stream);
  using Step = decltype(&bde_svgd_step_small_sgd);
  const int rc = reinterpret_cast<Step>(e_step)(a.P, a.G, a.s0, a.m, d, a.ld, static_cast<float>(l2_reg),
                                                static_cast<float>(kernel_grad_scale), static_cast<float>(dataset_size), lr,
                                                momentum, dampening, weight_decay, nesterov ? 1 : 0, first ? 1 : 0, a.ws,
                                                a.kstat, reinterpret_cast<void*>(stream));
  TORCH_CHECK(rc == 0, "bde_svgd_step_small_sgd failed with code ", rc);
}

void small_step_adam(int64_t e_gather, int64_t e_step, at::Tensor seg_ptrs, at::Tensor chunks, at::Tensor P, at::Tensor G,
                     at::Tensor exp_avg, at::Tensor exp_avg_sq, at::Tensor ws, at::Tensor kstat, int64_t d, double l2_reg,
                     double kernel_grad_scale, double dataset_size, double lr, double beta1, double beta2, double eps,
                     double weight_decay, int64_t step0, int64_t stream) {
  const SmallStepArgs a = small_step_common(e_gather, seg_ptrs, chunks, P, G, exp_avg, exp_avg_sq, ws, kstat, d, stream);
  using Step = decltype(&bde_svgd_step_small_adam);
  const int rc = reinterpret_cast<Step>(e_step)(a.P, a.G, a.s0, a.s1, a.m, d, a.ld, static_cast<float>(l2_reg),
                                                static_cast<float>(kernel_grad_scale), static_cast<float>(dataset_size), lr,
                                                beta1, beta2, eps, weight_decay, step0, a.ws, a.kstat,
                                                reinterpret_cast<void*>(stream));
  TORCH_CHECK(rc == 0, "bde_svgd_step_small_adam failed with code ", rc);
}

}  // namespace

void bind_autograd_nodes(py::module_& m);   // host_autograd.cpp: the Bayesian layers' autograd nodes

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  bind_autograd_nodes(m);
  py::class_<Layout>(m, "Layout")
      .def(py::init<std::vector<int64_t>, std::vector<int64_t>, std::vector<std::vector<int64_t>>>())
      .def("views", &Layout::views, "per-parameter views of a flat row")
      .def("point_data", &Layout::point_data, "param.data = its view of the row, for every parameter");
  py::class_<ParticleSet>(m, "ParticleSet")
      .def(py::init<std::vector<at::Tensor>, std::vector<std::vector<at::Tensor>>, std::vector<std::vector<at::Tensor>>>())
      .def("use", &ParticleSet::use)
      .def("begin", &ParticleSet::begin)
      .def("set_grads", &ParticleSet::set_grads)
      .def("end", &ParticleSet::end)
      .def("end_begin", &ParticleSet::end_begin)
      .def("release", &ParticleSet::release);
  m.def("repoint", &repoint, "param.data / param.grad = views, for whole parameter lists", py::arg("params"),
        py::arg("datas"), py::arg("grads"));
  m.def("mean_losses", &mean_losses, "sum of fp32 scalar tensors / divisor by one bde_mean_scalars call; False: not applicable",
        py::arg("losses"), py::arg("out"), py::arg("divisor"), py::arg("entry"), py::arg("stream"));
  m.def("small_step_sgd", &small_step_sgd, "bde_svgd_gather_seg + bde_svgd_step_small_sgd on the optimizer's buffers");
  m.def("small_step_adam", &small_step_adam, "bde_svgd_gather_seg + bde_svgd_step_small_adam on the optimizer's buffers");
  m.def("clear_grads", &clear_grads, "param.grad = None for a whole parameter list");
  m.def("collect_grads", &collect_grads, "record where the gradients live (no copy); returns the tensors taken by reference",
        py::arg("params"), py::arg("views"), py::arg("table"), py::arg("j"), py::arg("M"), py::arg("zero_addr") = 0,
        py::arg("take") = false);
  m.def("adopt_grads", &adopt_grads, "gradients -> flat views (multi-tensor copy/add), views become .grad", py::arg("params"),
        py::arg("views"), py::arg("add"));
}
