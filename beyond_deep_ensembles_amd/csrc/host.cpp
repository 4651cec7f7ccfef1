// Native host helper of the optimizer shells: the per-tensor pointer-aliasing loops in C++ (the layers' autograd
// nodes live in host_autograd.cpp, same module).
//
// The reference re-points `param.data` (and hands over `param.grad`) tensor by tensor in Python
// (src/algos/svgd.py:93-96,120-127; swag.py:58,81; ivorn.py:111).  With ResNet-50's 161 tensors an
// SVGD step does that 2 * 8 * 161 times -- ~4 ms of interpreter time around 0.5 ms of kernels.  Here
// the same loop runs over the already-unpacked tensor lists.  No arithmetic, no device work: plumbing.
#include <torch/extension.h>

#include <vector>

namespace {

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
    for (size_t i = 0; i < offsets_.size(); ++i) {
      at::Tensor p = params[i];
      p.set_data(row.narrow(0, offsets_[i], numels_[i]).view(shapes_[i]));
    }
  }

 private:
  std::vector<int64_t> offsets_, numels_;
  std::vector<std::vector<int64_t>> shapes_;
};

}  // namespace

void bind_autograd_nodes(py::module_& m);   // host_autograd.cpp: the Bayesian layers' autograd nodes

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  bind_autograd_nodes(m);
  py::class_<Layout>(m, "Layout")
      .def(py::init<std::vector<int64_t>, std::vector<int64_t>, std::vector<std::vector<int64_t>>>())
      .def("views", &Layout::views, "per-parameter views of a flat row")
      .def("point_data", &Layout::point_data, "param.data = its view of the row, for every parameter");
  m.def("repoint", &repoint, "param.data / param.grad = views, for whole parameter lists", py::arg("params"),
        py::arg("datas"), py::arg("grads"));
  m.def("clear_grads", &clear_grads, "param.grad = None for a whole parameter list");
  m.def("adopt_grads", &adopt_grads, "gradients -> flat views (multi-tensor copy/add), views become .grad", py::arg("params"),
        py::arg("views"), py::arg("add"));
}
