// gendr_torch.cpp -- the autograd node of the generalized soft rasterizer in C++ (PyTorch-ROCm), over the C ABI of libgendr_hip.so.
//
// The reference's host path is one C++ call per pass (gendr/cuda/generalized_renderer_cuda.cpp:74-192: pybind -> launcher); its
// callers are eager Python loops.  Through the ctypes binding an eager step of this build costs the host ~100 us of Python frames
// (autograd.Function.apply, ctypes marshalling, six tensor constructors) -- with a 0.23-ms GPU step at BASELINE config 2 that left
// eager launches 2-8 % behind a replayed HIP graph depending on the box.  This module is the same host logic
// as gendr_amd/functional/renderer.py (GenDRFunction.forward / backward) as a torch::autograd::Function: two native calls per step
// and nothing else.  It holds NO kernels and links NO libgendr_hip.so: the Python layer hands it the addresses of the C-ABI entry
// points of the library variant it loaded (bind()), so the boundary stays the C ABI of include/gendr_hip.h.
//
// Plumbing, not product: device memory and the stream come from PyTorch (c10::hip, its CUDA-masquerading guard and stream types), everything that computes is behind the ABI.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>      // PyTorch-ROCm tensors carry DeviceType::CUDA: its own guard / stream types
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/gendr_hip.h"

namespace {

typedef int (*validate_fn)(const gendr_params*, int, int, int);
typedef unsigned long long (*ws_bytes_fn)(int, int, int, const gendr_params*);
typedef int (*forward_fn)(const float*, const float*, float*, float*, void*, int, int, int, const gendr_params*, void*);
typedef int (*backward_fn)(const float*, const float*, const float*, const float*, const void*, const float*, float*, float*,
                           int, int, int, const gendr_params*, void*);
typedef const char* (*errstr_fn)(int);

struct Api { validate_fn validate; ws_bytes_fn ws_bytes; forward_fn forward; backward_fn backward; errstr_fn errstr; };
std::vector<Api> g_api;      // one per bound library variant (gendr_amd/_native.py: default, exact, ...)
// what the last calls did (tests/test_gpu_api.py: a differentiated step must run with pair hints and reuse the cleared buffer)
int64_t g_forward_with_grad = 0, g_forward_with_hints_off = 0, g_backward_prefilled = 0, g_backward_filled_here = 0;

void check(const Api& api, int code, const char* what)
{
    if (code == GENDR_OK) return;
    const std::string msg = std::string(what) + ": " + api.errstr(code) + " (code " + std::to_string(code) + ")";
    if (code <= GENDR_E_SHAPE && code >= GENDR_E_TCONORM_PARAM) throw py::value_error(msg);     // as functional/renderer.py check()
    throw std::runtime_error(msg);
}

struct GenDRNode : public torch::autograd::Function<GenDRNode> {
    static torch::Tensor forward(torch::autograd::AutogradContext* ctx, const torch::Tensor& face_vertices, const torch::Tensor& textures,
                                 const std::string& params_bytes, int64_t api_slot, bool fused_clear, bool want_grad)
    {
        const Api& api = g_api.at((size_t)api_slot);
        gendr_params p;
        TORCH_CHECK(params_bytes.size() == sizeof(gendr_params), "gendr_params layout mismatch");
        std::memcpy(&p, params_bytes.data(), sizeof(p));
        if (!face_vertices.is_cuda()) throw py::type_error("GenDR only supports CUDA Tensors (face_vertices is on " + face_vertices.device().str() + ").");
        if (!textures.is_cuda()) throw py::type_error("GenDR only supports CUDA Tensors (textures is on " + textures.device().str() + ").");
        TORCH_CHECK(face_vertices.dim() >= 3, "face_vertices must be [B, nf, 3, 3] or [B, nf, 9]");
        const int64_t B = face_vertices.size(0), nf = face_vertices.size(1);
        torch::Tensor faces = face_vertices.detach().reshape({B, nf, 9}).to(torch::kFloat32).contiguous();
        torch::Tensor tex = textures.detach().to(faces.device(), torch::kFloat32).contiguous();
        if (tex.dim() != 4 || tex.size(0) != B || tex.size(1) != nf || tex.size(3) != 3)
            throw py::value_error("textures must be [B, nf, T, 3] matching face_vertices [B, nf, 3, 3]");
        const int64_t T = tex.size(2);
        check(api, api.validate(&p, (int)B, (int)nf, (int)T), "gendr.render");
        // (`want_grad` comes from render(): inside a custom Function's forward the grad mode is already switched off, and without a
        // differentiable input the node has no edges for ctx->needs_input_grad() to ask -- a first version asked GradMode here, got
        // "off" every time, and ran every step without pair hints and without the fused gradient clear: backward 100 instead of 88 us)
        // pair hints (ABI 6) are for the backward call: a forward pass nobody differentiates does not pay for them
        if (!want_grad && p.pair_hints == 0) p.pair_hints = -1;
        g_forward_with_grad += want_grad;
        g_forward_with_hints_off += p.pair_hints < 0;
        const auto opts = faces.options();
        const int64_t isz = p.image_size;
        torch::Tensor rgba = torch::empty({B, 4, isz, isz}, opts), aux = torch::empty({B, 2, isz, isz}, opts);
        const unsigned long long nbytes = api.ws_bytes((int)B, (int)nf, (int)T, &p);
        torch::Tensor ws = torch::empty({(int64_t)std::max<unsigned long long>(nbytes, 256)}, opts.dtype(torch::kUInt8));
        // the gradients of the coming backward call: allocated now, zero-filled by the setup stage of gendr_forward on its way
        // (gendr_params.clear_ptr); used once
        torch::Tensor flat;
        const int64_t n_f = B * nf * 9, n_t = tex.numel(), n_f_pad = (n_f + 63) / 64 * 64;
        if (want_grad && fused_clear && B * nf > 0) {
            flat = torch::empty({(n_f_pad + n_t + 3) / 4 * 4}, opts);
            p.clear_ptr = flat.data_ptr();
            p.clear_floats = (unsigned long long)flat.numel();
        }
        {
            c10::hip::HIPGuardMasqueradingAsCUDA guard(faces.device());
            void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(faces.device().index()).stream();
            const int rc = api.forward(faces.data_ptr<float>(), tex.data_ptr<float>(), rgba.data_ptr<float>(), aux.data_ptr<float>(),
                                       ws.data_ptr(), (int)B, (int)nf, (int)T, &p, stream);
            p.clear_ptr = nullptr;
            p.clear_floats = 0;
            check(api, rc, "gendr_forward");
        }
        ctx->save_for_backward({faces, tex, rgba, ws, aux});
        ctx->saved_data["params"] = std::string(reinterpret_cast<const char*>(&p), sizeof(p));
        ctx->saved_data["api"] = api_slot;
        ctx->saved_data["fv_sizes"] = face_vertices.sizes().vec();
        ctx->saved_data["fv_dtype"] = (int64_t)face_vertices.scalar_type();
        ctx->saved_data["tex_dtype"] = (int64_t)textures.scalar_type();
        if (flat.defined()) ctx->saved_data["flat"] = flat;
        return rgba;
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grad_out)
    {
        const auto saved = ctx->get_saved_variables();
        const torch::Tensor &faces = saved[0], &tex = saved[1], &rgba = saved[2], &ws = saved[3], &aux = saved[4];
        const Api& api = g_api.at((size_t)ctx->saved_data["api"].toInt());
        gendr_params p;
        const std::string pb = ctx->saved_data["params"].toStringRef();
        std::memcpy(&p, pb.data(), sizeof(p));
        const int64_t B = faces.size(0), nf = faces.size(1), T = tex.size(2);
        // like GenDRFunction (@once_differentiable): the gradients returned here carry no graph, so a double backward
        // (create_graph=True) must fail loudly instead of treating them as constants
        TORCH_CHECK(!(at::GradMode::is_enabled() && grad_out[0].defined() && grad_out[0].requires_grad()),
                    "gendr render: the backward pass is hand-derived and not differentiable a second time (create_graph=True is not supported)");
        torch::Tensor grad = grad_out[0].to(faces.device(), torch::kFloat32).contiguous();
        const int64_t n_f = B * nf * 9, n_t = tex.numel(), n_f_pad = (n_f + 63) / 64 * 64;
        torch::Tensor flat;
        auto it = ctx->saved_data.find("flat");
        if (it != ctx->saved_data.end() && it->second.isTensor()) {
            flat = it->second.toTensor();
            g_backward_prefilled++;
            ctx->saved_data.erase("flat");               // cleared by the forward call: good for one use (retain_graph: a fresh, filled one)
        } else {
            flat = torch::zeros({(n_f_pad + n_t + 3) / 4 * 4}, faces.options());
            g_backward_filled_here++;
        }
        torch::Tensor grad_faces = flat.narrow(0, 0, n_f).view({B, nf, 9});
        torch::Tensor grad_tex = flat.narrow(0, n_f_pad, n_t).view(tex.sizes());
        if (B * nf > 0) {
            c10::hip::HIPGuardMasqueradingAsCUDA guard(faces.device());
            void* stream = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(faces.device().index()).stream();
            check(api, api.backward(faces.data_ptr<float>(), tex.data_ptr<float>(), rgba.data_ptr<float>(), aux.data_ptr<float>(), ws.data_ptr(),
                                    grad.data_ptr<float>(), grad_faces.data_ptr<float>(), grad_tex.data_ptr<float>(),
                                    (int)B, (int)nf, (int)T, &p, stream), "gendr_backward");
        }
        const auto fv_sizes = ctx->saved_data["fv_sizes"].toIntVector();
        torch::Tensor gf = grad_faces.reshape(fv_sizes).to((c10::ScalarType)ctx->saved_data["fv_dtype"].toInt());
        torch::Tensor gt = grad_tex.to((c10::ScalarType)ctx->saved_data["tex_dtype"].toInt());
        return {gf, gt, torch::Tensor(), torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

int64_t bind(int64_t validate, int64_t ws_bytes, int64_t forward, int64_t backward, int64_t errstr, int64_t params_size, int64_t abi)
{
    TORCH_CHECK(params_size == (int64_t)sizeof(gendr_params) && abi == GENDR_ABI_VERSION, "gendr_torch: built against another ABI of include/gendr_hip.h");
    g_api.push_back(Api{(validate_fn)validate, (ws_bytes_fn)ws_bytes, (forward_fn)forward, (backward_fn)backward, (errstr_fn)errstr});
    return (int64_t)g_api.size() - 1;
}

torch::Tensor render(const torch::Tensor& face_vertices, const torch::Tensor& textures, const py::bytes& params, int64_t api_slot, bool fused_clear)
{
    const bool want_grad = at::GradMode::is_enabled() && (face_vertices.requires_grad() || textures.requires_grad());
    return GenDRNode::apply(face_vertices, textures, std::string(params), api_slot, fused_clear, want_grad);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("bind", &bind, "registers the C-ABI entry points of one loaded libgendr_hip*.so; returns its slot");
    m.def("stats", []() { return py::dict(py::arg("forward_with_grad") = g_forward_with_grad, py::arg("forward_with_hints_off") = g_forward_with_hints_off,
                                          py::arg("backward_prefilled") = g_backward_prefilled, py::arg("backward_filled_here") = g_backward_filled_here); },
          "counters of what the node's calls did so far");
    m.def("render", &render, "GenDRFunction as a C++ autograd node: (face_vertices, textures, gendr_params bytes, api slot, fused clear) -> rgba");
}
