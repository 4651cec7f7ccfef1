// Autograd nodes of the Bayesian layers in C++ (part of lib/_bde_host.so).
//
// The layers' fused ops (bde_lrt_linear_fwd/bwd, bde_local_reparam_fwd/bwd, bde_var_operand_fwd/bwd) are single
// launches of 10-40 us; wrapped as Python torch.autograd.Function they cost 40-55 us of interpreter time per node and
// direction, more than the ATen nodes they replace below a few million elements (tools/lrt_host_profile.py,
// tools/conv_layer_bench.py).  The same nodes here: argument checks, output allocation and ONE call through the C ABI
// on torch's current stream.  No arithmetic lives in this file; the Python Functions in bbb_layers.py remain the
// fallback when the helper is not built (same kernels, same results).
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>

#include "../../include/bde_hip.h"

namespace {

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void* current_stream(const at::Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_f32_cuda(const at::Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat, name, ": expected a float32 CUDA (HIP) tensor");
}
// the kernels produce plain tensors: a second differentiation through a node would silently see constants
void check_once_differentiable(const variable_list& grad_outputs) {
  TORCH_CHECK(!(at::GradMode::is_enabled() && grad_outputs[0].defined() && grad_outputs[0].requires_grad()),
              "the fused layer ops are differentiable once (no double backward)");
}
const float* ptr(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* mptr(at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
at::Tensor opt(const c10::optional<at::Tensor>& t) { return t.has_value() ? *t : at::Tensor(); }

// ---- BBBLinear, sampling="activations" (bbb_layers.py:61-80): forward bde_lrt_linear_fwd, backward bde_lrt_linear_bwd
struct LrtLinear : public torch::autograd::Function<LrtLinear> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w_mu, const at::Tensor& w_rho,
                            const c10::optional<at::Tensor>& b_mu_, const c10::optional<at::Tensor>& b_rho_,
                            bool clamp_bias, const c10::optional<at::Tensor>& eps_, int64_t seed, int64_t stream_id,
                            const c10::optional<at::Tensor>& w_s2_, const c10::optional<at::Tensor>& w_ds2_) {
    at::Tensor b_mu = opt(b_mu_), b_rho = opt(b_rho_), eps = opt(eps_);
    // sigma^2 / its rho-derivative of THIS version of w_rho (bde_lrt_sigma_cache), or undefined: on the fly
    at::Tensor w_s2 = opt(w_s2_), w_ds2 = opt(w_ds2_);
    check_f32_cuda(x, "x");
    check_f32_cuda(w_mu, "w_mu");
    check_f32_cuda(w_rho, "w_rho");
    TORCH_CHECK(b_mu.defined() == b_rho.defined(), "bias mean and rho come together");
    c10::DeviceGuard guard(x.device());
    at::Tensor x2d = x.reshape({-1, x.size(-1)});
    if (x2d.stride(-1) != 1) x2d = x2d.contiguous();
    const at::Tensor wm = w_mu.detach().contiguous(), wr = w_rho.detach().contiguous();
    const int B = static_cast<int>(x2d.size(0)), I = static_cast<int>(x2d.size(1)), O = static_cast<int>(wm.size(0));
    const size_t ws_bytes = bde_lrt_linear_ws_bytes(B, I, O);
    TORCH_CHECK(ws_bytes > 0, "lrt_linear: unsupported shape B=", B, ", I=", I, ", O=", O);
    at::Tensor out = at::empty({B, O}, x2d.options()), var = at::empty({B, O}, x2d.options());
    at::Tensor ws = at::empty({static_cast<int64_t>(ws_bytes / 4)}, x2d.options());
    at::Tensor e;
    if (eps.defined()) e = eps.reshape({B, O}).contiguous();
    const at::Tensor xd = x2d.detach();
    const int rc = bde_lrt_linear_fwd(ptr(xd), xd.stride(0), ptr(wm), ptr(wr), ptr(w_s2), ptr(b_mu), ptr(b_rho), clamp_bias ? 1 : 0,
                                      ptr(e), static_cast<uint64_t>(seed), static_cast<uint64_t>(stream_id), mptr(out),
                                      mptr(var), B, I, O, ws.data_ptr(), current_stream(x2d));
    TORCH_CHECK(rc == 0, "bde_lrt_linear_fwd failed with code ", rc);
    // the cache tensors of THIS weight version are saved like any other operand (a refresh writes new tensors, so a
    // node always sees the cache of the version its forward ran on; editing rho before the backward trips autograd's
    // version check on w_rho as it always did)
    ctx->save_for_backward({x2d, w_mu, w_rho, b_rho, var, e, w_s2, w_ds2});
    ctx->saved_data["clamp_bias"] = clamp_bias;
    ctx->saved_data["seed"] = seed;
    ctx->saved_data["stream_id"] = stream_id;
    ctx->saved_data["x_shape"] = x.sizes().vec();
    ctx->saved_data["has_bias"] = b_rho.defined();
    std::vector<int64_t> out_shape = x.sizes().vec();
    out_shape.back() = O;
    return out.view(out_shape);
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
    check_once_differentiable(grad_outputs);
    const variable_list saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &w_mu = saved[1], &w_rho = saved[2], &b_rho = saved[3], &var = saved[4], &eps = saved[5];
    const at::Tensor &w_s2 = saved[6], &w_ds2 = saved[7];
    const bool has_bias = ctx->saved_data["has_bias"].toBool();
    c10::DeviceGuard guard(x.device());
    const at::Tensor g = grad_outputs[0].reshape(var.sizes()).contiguous();
    const int B = static_cast<int>(x.size(0)), I = static_cast<int>(x.size(1)), O = static_cast<int>(w_mu.size(0));
    const at::Tensor wm = w_mu.detach().contiguous(), wr = w_rho.detach().contiguous();
    at::Tensor g_x;
    if (ctx->needs_input_grad(0)) g_x = at::empty({B, I}, var.options());
    at::Tensor g_wmu = at::empty_like(wm), g_wrho = at::empty_like(wr), g_bmu, g_brho, br;
    if (has_bias) {
      br = b_rho.detach();
      g_bmu = at::empty_like(br);
      g_brho = at::empty_like(br);
    }
    at::Tensor ws = at::empty({static_cast<int64_t>(bde_lrt_linear_bwd_ws_bytes(B, I, O) / 4)}, var.options());
    const at::Tensor xd = x.detach();
    const int rc = bde_lrt_linear_bwd(ptr(xd), xd.stride(0), ptr(wm), ptr(wr), ptr(w_s2), ptr(w_ds2), ptr(br),
                                      ctx->saved_data["clamp_bias"].toBool() ? 1 : 0,
                                      ptr(g), ptr(var), ptr(eps), static_cast<uint64_t>(ctx->saved_data["seed"].toInt()),
                                      static_cast<uint64_t>(ctx->saved_data["stream_id"].toInt()), mptr(g_x), mptr(g_wmu),
                                      mptr(g_wrho), mptr(g_bmu), mptr(g_brho), B, I, O, ws.data_ptr(), current_stream(x));
    TORCH_CHECK(rc == 0, "bde_lrt_linear_bwd failed with code ", rc);
    if (g_x.defined()) g_x = g_x.view(ctx->saved_data["x_shape"].toIntVector());
    return {g_x, g_wmu, g_wrho, g_bmu, g_brho, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

// ---- epilogue out = mean + sqrt(var) * eps (bbb_layers.py:70-80): bde_local_reparam_fwd / bwd
struct LocalReparam : public torch::autograd::Function<LocalReparam> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& mean, const at::Tensor& var,
                            const c10::optional<at::Tensor>& eps_, int64_t seed, int64_t stream_id) {
    check_f32_cuda(mean, "mean");
    check_f32_cuda(var, "var");
    c10::DeviceGuard guard(mean.device());
    const at::Tensor m = mean.detach().contiguous(), v = var.detach().contiguous();
    at::Tensor e = opt(eps_);
    if (e.defined()) e = e.contiguous();
    at::Tensor out = at::empty_like(m);
    const int rc = bde_local_reparam_fwd(ptr(m), ptr(v), ptr(e), static_cast<uint64_t>(seed), static_cast<uint64_t>(stream_id),
                                         mptr(out), m.numel(), current_stream(m));
    TORCH_CHECK(rc == 0, "bde_local_reparam_fwd failed with code ", rc);
    ctx->save_for_backward({v, e});
    ctx->saved_data["seed"] = seed;
    ctx->saved_data["stream_id"] = stream_id;
    return out.view(mean.sizes());
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
    check_once_differentiable(grad_outputs);
    const variable_list saved = ctx->get_saved_variables();
    const at::Tensor &v = saved[0], &e = saved[1];
    c10::DeviceGuard guard(v.device());
    const at::Tensor g = grad_outputs[0].contiguous();
    at::Tensor gvar = at::empty_like(v);
    const int rc = bde_local_reparam_bwd(ptr(g), ptr(v), ptr(e), static_cast<uint64_t>(ctx->saved_data["seed"].toInt()),
                                         static_cast<uint64_t>(ctx->saved_data["stream_id"].toInt()), mptr(gvar), v.numel(),
                                         current_stream(v));
    TORCH_CHECK(rc == 0, "bde_local_reparam_bwd failed with code ", rc);
    return {grad_outputs[0], gvar.view(grad_outputs[0].sizes()), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

// ---- operands of the variance product (bbb_layers.py:66-67,71,150-153): bde_var_operand_fwd / bwd
struct VarOperand : public torch::autograd::Function<VarOperand> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& v, int64_t mode) {
    check_f32_cuda(v, "v");
    c10::DeviceGuard guard(v.device());
    const at::Tensor vc = v.detach().contiguous();
    at::Tensor out = at::empty_like(vc);
    const int rc = bde_var_operand_fwd(ptr(vc), static_cast<int>(mode), mptr(out), vc.numel(), current_stream(vc));
    TORCH_CHECK(rc == 0, "bde_var_operand_fwd failed with code ", rc);
    ctx->save_for_backward({vc});
    ctx->saved_data["mode"] = mode;
    return out.view(v.sizes());
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
    check_once_differentiable(grad_outputs);
    const at::Tensor vc = ctx->get_saved_variables()[0];
    c10::DeviceGuard guard(vc.device());
    const at::Tensor g = grad_outputs[0].contiguous();
    at::Tensor gv = at::empty_like(vc);
    const int rc = bde_var_operand_bwd(ptr(g), ptr(vc), static_cast<int>(ctx->saved_data["mode"].toInt()), mptr(gv),
                                       vc.numel(), current_stream(vc));
    TORCH_CHECK(rc == 0, "bde_var_operand_bwd failed with code ", rc);
    return {gv.view(grad_outputs[0].sizes()), at::Tensor()};
  }
};

// ---- BBBConv2d, sampling="activations" (bbb_layers.py:146-154): forward bde_conv_lrt_fwd (both convolutions + the sampling
// epilogue in one launch); backward bde_local_reparam_bwd (g_var) + bde_conv_lrt_bwd_data + bde_conv_lrt_bwd_weight + two
// channel sums for the bias.  `wbuf` = the weight buffer bde_conv_lrt_prep filled for THIS version of (w_mu, w_rho, b_rho).
struct ConvLrt : public torch::autograd::Function<ConvLrt> {
  static at::Tensor forward(AutogradContext* ctx, const at::Tensor& x, const at::Tensor& w_mu, const at::Tensor& w_rho,
                            const c10::optional<at::Tensor>& b_mu_, const c10::optional<at::Tensor>& b_rho_, int64_t sh,
                            int64_t sw, int64_t ph, int64_t pw, const c10::optional<at::Tensor>& eps_, int64_t seed,
                            int64_t stream_id, const at::Tensor& wbuf, bool phases, bool want_var) {
    at::Tensor b_mu = opt(b_mu_), b_rho = opt(b_rho_), eps = opt(eps_);
    check_f32_cuda(x, "x");
    check_f32_cuda(w_mu, "w_mu");
    check_f32_cuda(w_rho, "w_rho");
    check_f32_cuda(wbuf, "wbuf");
    TORCH_CHECK(x.dim() == 4 && w_mu.dim() == 4 && x.size(1) == w_mu.size(1), "conv_lrt: x [N, C, H, W], w [O, C, KH, KW]");
    TORCH_CHECK(b_mu.defined() == b_rho.defined(), "bias mean and rho come together");
    c10::DeviceGuard guard(x.device());
    const at::Tensor xc = x.detach().contiguous();
    const int N = static_cast<int>(xc.size(0)), C = static_cast<int>(xc.size(1)), H = static_cast<int>(xc.size(2)),
              W = static_cast<int>(xc.size(3)), O = static_cast<int>(w_mu.size(0)), KH = static_cast<int>(w_mu.size(2)),
              KW = static_cast<int>(w_mu.size(3));
    TORCH_CHECK(w_rho.sizes() == w_mu.sizes() && wbuf.is_contiguous() &&
                    static_cast<size_t>(wbuf.numel()) >= bde_conv_lrt_prep_floats(O, C, KH, KW),
                "conv_lrt: wbuf does not belong to a layer of this shape");
    const int64_t Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
    TORCH_CHECK(Ho >= 1 && Wo >= 1, "conv_lrt: empty output");
    // the total variance is what the backward needs: a forward nobody will differentiate (want_var false) does not write it
    at::Tensor out = at::empty({N, O, Ho, Wo}, xc.options()), var;
    if (want_var) var = at::empty({N, O, Ho, Wo}, xc.options());
    at::Tensor e;
    if (eps.defined()) e = eps.reshape(out.sizes()).contiguous();
    at::Tensor bm;
    if (b_mu.defined()) bm = b_mu.detach().contiguous();
    const int rc = bde_conv_lrt_fwd(ptr(xc), ptr(wbuf), ptr(bm), b_rho.defined() ? 1 : 0, ptr(e), static_cast<uint64_t>(seed),
                                    static_cast<uint64_t>(stream_id), mptr(out), mptr(var), N, C, H, W, O, KH, KW,
                                    static_cast<int>(sh), static_cast<int>(sw), static_cast<int>(ph), static_cast<int>(pw),
                                    current_stream(xc));
    TORCH_CHECK(rc == 0, "bde_conv_lrt_fwd failed with code ", rc);
    ctx->save_for_backward({xc, w_mu, w_rho, b_rho, var, e, wbuf});
    ctx->saved_data["geo"] = std::vector<int64_t>{sh, sw, ph, pw};
    ctx->saved_data["seed"] = seed;
    ctx->saved_data["stream_id"] = stream_id;
    ctx->saved_data["phases"] = phases;          // wbuf carries the per-phase input-gradient matrices of this stride / padding
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list grad_outputs) {
    check_once_differentiable(grad_outputs);
    const variable_list saved = ctx->get_saved_variables();
    const at::Tensor &x = saved[0], &w_mu = saved[1], &w_rho = saved[2], &b_rho = saved[3], &var = saved[4], &eps = saved[5],
                     &wbuf = saved[6];
    const std::vector<int64_t> geo = ctx->saved_data["geo"].toIntVector();
    const int sh = static_cast<int>(geo[0]), sw = static_cast<int>(geo[1]), ph = static_cast<int>(geo[2]), pw = static_cast<int>(geo[3]);
    c10::DeviceGuard guard(x.device());
    const int N = static_cast<int>(x.size(0)), C = static_cast<int>(x.size(1)), H = static_cast<int>(x.size(2)),
              W = static_cast<int>(x.size(3)), O = static_cast<int>(w_mu.size(0)), KH = static_cast<int>(w_mu.size(2)),
              KW = static_cast<int>(w_mu.size(3));
    const at::Tensor g = grad_outputs[0].contiguous();
    at::Tensor gvar = at::empty_like(g);
    void* stream = current_stream(x);
    // ONE pass: g_var and (with a bias) both bias gradients (channel sums of g / g_var + the rho chain rule in the finish)
    at::Tensor g_bmu, g_brho, br, gws;
    if (b_rho.defined()) {
      br = b_rho.detach().contiguous();
      g_bmu = at::empty_like(br);
      g_brho = at::empty_like(br);
      gws = at::empty({static_cast<int64_t>(bde_conv_lrt_gvar_ws_bytes(N, O) / 8)}, x.options().dtype(at::kDouble));
    }
    int rc = bde_conv_lrt_gvar_bias(ptr(g), ptr(var), ptr(eps), static_cast<uint64_t>(ctx->saved_data["seed"].toInt()),
                                    static_cast<uint64_t>(ctx->saved_data["stream_id"].toInt()), mptr(gvar),
                                    br.defined() ? ptr(br) : nullptr, g_bmu.defined() ? mptr(g_bmu) : nullptr,
                                    g_brho.defined() ? mptr(g_brho) : nullptr, gws.defined() ? gws.data_ptr() : nullptr, N, O,
                                    g.size(2) * g.size(3), stream);
    TORCH_CHECK(rc == 0, "bde_conv_lrt_gvar_bias failed with code ", rc);
    at::Tensor g_x;
    if (ctx->needs_input_grad(0)) {
      g_x = at::empty_like(x);
      rc = (ctx->saved_data["phases"].toBool() ? bde_conv_lrt_bwd_data_phases : bde_conv_lrt_bwd_data)(
          ptr(g), ptr(gvar), ptr(wbuf), ptr(x), mptr(g_x), N, C, H, W, O, KH, KW, sh, sw, ph, pw, stream);
      TORCH_CHECK(rc == 0, "bde_conv_lrt_bwd_data failed with code ", rc);
    }
    const at::Tensor wr = w_rho.detach().contiguous();
    at::Tensor g_wmu = at::empty_like(wr), g_wrho = at::empty_like(wr);
    const size_t ws_bytes = bde_conv_lrt_bwd_weight_ws_bytes(N, C, H, W, O, KH, KW, sh, sw, ph, pw);
    TORCH_CHECK(ws_bytes > 0, "conv_lrt: unsupported geometry in the weight-gradient pass");
    at::Tensor ws = at::empty({static_cast<int64_t>((ws_bytes + 3) / 4)}, x.options());
    rc = bde_conv_lrt_bwd_weight(ptr(x), ptr(g), ptr(gvar), ptr(wr), ws.data_ptr(), static_cast<size_t>(ws.nbytes()), mptr(g_wmu),
                                 mptr(g_wrho), N, C, H, W, O, KH, KW, sh, sw, ph, pw, stream);
    TORCH_CHECK(rc == 0, "bde_conv_lrt_bwd_weight failed with code ", rc);
    return {g_x, g_wmu, g_wrho, g_bmu, g_brho, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
            at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
  }
};

}  // namespace

void bind_autograd_nodes(py::module_& m) {
  m.def("lrt_linear",
        [](const at::Tensor& x, const at::Tensor& w_mu, const at::Tensor& w_rho, const c10::optional<at::Tensor>& b_mu,
           const c10::optional<at::Tensor>& b_rho, bool clamp_bias, const c10::optional<at::Tensor>& eps, int64_t seed,
           int64_t stream_id, const c10::optional<at::Tensor>& w_s2, const c10::optional<at::Tensor>& w_ds2) {
          return LrtLinear::apply(x, w_mu, w_rho, b_mu, b_rho, clamp_bias, eps, seed, stream_id, w_s2, w_ds2);
        },
        "BBBLinear forward (local reparameterisation) with its fused backward", py::arg("x"), py::arg("w_mu"),
        py::arg("w_rho"), py::arg("b_mu"), py::arg("b_rho"), py::arg("clamp_bias"), py::arg("eps"), py::arg("seed"),
        py::arg("stream_id"), py::arg("w_s2") = py::none(), py::arg("w_ds2") = py::none());
  m.def("local_reparam",
        [](const at::Tensor& mean, const at::Tensor& var, const c10::optional<at::Tensor>& eps, int64_t seed,
           int64_t stream_id) { return LocalReparam::apply(mean, var, eps, seed, stream_id); },
        "mean + sqrt(var) * eps", py::arg("mean"), py::arg("var"), py::arg("eps"), py::arg("seed"), py::arg("stream_id"));
  m.def("conv_lrt",
        [](const at::Tensor& x, const at::Tensor& w_mu, const at::Tensor& w_rho, const c10::optional<at::Tensor>& b_mu,
           const c10::optional<at::Tensor>& b_rho, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
           const c10::optional<at::Tensor>& eps, int64_t seed, int64_t stream_id, const at::Tensor& wbuf, bool phases,
           bool want_var) {
          return ConvLrt::apply(x, w_mu, w_rho, b_mu, b_rho, sh, sw, ph, pw, eps, seed, stream_id, wbuf, phases, want_var);
        },
        "BBBConv2d forward (local reparameterisation, fused) with its fused backward", py::arg("x"), py::arg("w_mu"),
        py::arg("w_rho"), py::arg("b_mu"), py::arg("b_rho"), py::arg("sh"), py::arg("sw"), py::arg("ph"), py::arg("pw"),
        py::arg("eps"), py::arg("seed"), py::arg("stream_id"), py::arg("wbuf"), py::arg("phases") = false,
        py::arg("want_var") = true);
  m.def("var_operand", [](const at::Tensor& v, int64_t mode) { return VarOperand::apply(v, mode); },
        "clamp(v^2) / clamp(softplus(v)^2) / softplus(v)^2", py::arg("v"), py::arg("mode"));
}
