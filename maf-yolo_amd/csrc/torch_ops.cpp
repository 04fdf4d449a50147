// PyTorch-ROCm custom-op layer over the C-ABI of libmafyolo_hip.so (BASELINE.json north_star: "exposed behind the
// reference's Model.forward() / non_max_suppression() surface as PyTorch-ROCm custom ops").
//
//   torch.ops.load_library("maf-yolo_amd/libmafyolo_torch.so")
//   y    = torch.ops.mafyolo.conv1x1_bias_act(x, w, b, act)            # Conv.forward_fuse (yolov6/layers/common.py:49-50), nn.Conv2d preds (:1331,1335)
//   y    = torch.ops.mafyolo.conv3x3s2_bias_act(x, w, b, act)          # RepVGGBlock deploy forward (:216-217), ConvWrapper (:76-83)
//   y    = torch.ops.mafyolo.dwconv_bias_act(x, w, b, act)             # UniRepLKNetBlock after reparameterize (:3085-3100)
//   pred = torch.ops.mafyolo.head_decode(cls, reg, strides)            # Detect_yaml.forward eval branch (yolov6/models/yolo.py:355-396)
//   rows, counts = torch.ops.mafyolo.decode_nms(pred, conf, iou, agnostic, multi_label, max_det, classes)   # non_max_suppression (yolov6/utils/nms.py:31-105)
//
// Every op takes / returns at::Tensor (NCHW shape, channels_last = NHWC memory, fp16 or fp32, on the HIP device), runs on the CURRENT HIP
// stream, allocates its outputs through the caching allocator, keeps no reference after it returns and reports errors as RuntimeError
// (TORCH_CHECK).  The backward pieces (data / weight gradients) are ops of their own; maf_yolo_amd/torch_ops.py wires them up with
// torch.library.register_autograd, registers the autocast rule (cast to the autocast dtype, like a convolution) and the fake (meta)
// kernels.  No kernel lives here: this file only marshals tensors into the C-ABI (include/mafyolo_hip.h).
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/mafyolo_hip.h"

namespace {

using at::Tensor;

void check(int rc, const char* what) { TORCH_CHECK(rc == 0, "mafyolo::", what, ": ", maf_last_error(), " (code ", rc, ")"); }

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

int dtype_of(const Tensor& t) {
    TORCH_CHECK(t.scalar_type() == at::kHalf || t.scalar_type() == at::kFloat, "mafyolo ops take fp16 or fp32 tensors, got ", t.scalar_type());
    return t.scalar_type() == at::kHalf ? MAF_F16 : MAF_F32;
}

Tensor nhwc(const Tensor& x) {
    TORCH_CHECK(x.is_cuda() && x.dim() == 4, "mafyolo ops take 4-d tensors on the HIP device (there is no CPU path)");
    return x.contiguous(at::MemoryFormat::ChannelsLast);
}

// (tile_p, tile_c) as maf-yolo_amd/pack.py:tile_for: least channel padding, then fewest tiles
std::pair<int, int> tile_for(int64_t cout, int64_t m, bool dgrad = false) {
    int best_ct = 2;
    int64_t best_pad = -1, best_tiles = -1;
    for (int ct : {8, 6, 4, 2}) {
        if (dgrad && ct == 6) continue;
        const int64_t tiles = (cout + 16 * ct - 1) / (16 * ct), pad = tiles * 16 * ct;
        if (best_pad < 0 || pad < best_pad || (pad == best_pad && tiles < best_tiles)) { best_pad = pad; best_tiles = tiles; best_ct = ct; }
    }
    const int pt = ((m + 127) / 128) * best_tiles >= 1024 ? 2 : 1;
    return {pt, best_ct};
}

Tensor padded_bias(const c10::optional<Tensor>& bias, int64_t cout, int ct, const Tensor& like) {
    const int64_t npad = (cout + 16 * ct - 1) / (16 * ct) * 16 * ct;
    Tensor b = at::zeros({npad}, like.options().dtype(at::kFloat).memory_format(c10::nullopt));
    if (bias.has_value() && bias->defined()) b.narrow(0, 0, cout).copy_(bias->to(at::kFloat));
    return b;
}

// fragment-packed [N, K] matrix (K = whole k-steps) on the device
Tensor pack_matrix(const Tensor& w2d, int64_t n, int64_t k, int transpose, int dt, int ct, const Tensor& like) {
    const int64_t bytes = maf_pack_w1x1_bytes((int)n, (int)k, transpose, dt, ct);
    Tensor buf = at::empty({bytes}, like.options().dtype(at::kByte).memory_format(c10::nullopt));
    Tensor wf = w2d.to(at::kFloat).contiguous();
    check(maf_pack_w1x1(wf.data_ptr<float>(), (int)n, (int)k, transpose, dt, ct, buf.data_ptr(), stream_of(like)), "pack_w1x1");
    return buf;
}

Tensor pack_3x3(const Tensor& w, bool transpose, int dt, int ct, const Tensor& like) {     // tap-major K, every tap padded to whole k-steps
    const int64_t ks = dt == MAF_F16 ? 32 : 16;
    Tensor m = transpose ? w.to(at::kFloat).permute({1, 2, 3, 0}) : w.to(at::kFloat).permute({0, 2, 3, 1});      // [N, 3, 3, K]
    const int64_t n = m.size(0), k = m.size(3), kp = (k + ks - 1) / ks * ks;
    Tensor big = at::constant_pad_nd(m, {0, kp - k}).reshape({n, 9 * kp}).contiguous();
    return pack_matrix(big, n, 9 * kp, 0, dt, ct, like);
}

void fill_src(maf_op_t& op, const Tensor& x, int mode = MAF_SRC_DIRECT) {
    op.nsrc = 1;
    op.src[0].ptr = x.data_ptr(); op.src[0].C = (int)x.size(1); op.src[0].stride = (int)x.stride(3); op.src[0].coff = 0; op.src[0].mode = mode;
}

Tensor conv_generic(const Tensor& x_in, const Tensor& wp, const Tensor& bias, int kind, int64_t cout, int64_t Ho, int64_t Wo, int64_t Hin, int64_t Win, int act,
                    int pt, int ct) {
    Tensor x = x_in;
    const int dt = dtype_of(x);
    Tensor out = at::empty({x.size(0), cout, Ho, Wo}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    maf_op_t op = {};
    op.kind = kind; op.dtype = dt; op.in_dtype = dt; op.act = act;
    op.B = (int)x.size(0); op.H = (int)Ho; op.W = (int)Wo; op.Hin = (int)Hin; op.Win = (int)Win; op.Cin = (int)x.size(1); op.Cout = (int)cout;
    fill_src(op, x);
    op.out = out.data_ptr(); op.out_stride = (int)out.stride(3); op.out_coff = 0;
    op.tile_p = pt; op.tile_c = ct;
    op.w = wp.data_ptr(); op.bias = bias.data_ptr<float>();
    check(maf_op_launch(&op, stream_of(x)), "conv launch");
    return out;
}

Tensor pad_channels8(const Tensor& x) {            // a 3-channel image for kernels that read 16-byte channel chunks
    const int64_t c = x.size(1), extra = (8 - c % 8) % 8;
    if (!extra) return x;
    return at::constant_pad_nd(x, {0, 0, 0, 0, 0, extra}).contiguous(at::MemoryFormat::ChannelsLast);
}

Tensor conv1x1_bias_act(const Tensor& x_in, const Tensor& w, const c10::optional<Tensor>& bias, int64_t act) {
    const c10::DeviceGuard device_guard(x_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    Tensor x = pad_channels8(nhwc(x_in));
    TORCH_CHECK(w.dim() == 4 && w.size(2) == 1 && w.size(3) == 1 && w.size(1) == x_in.size(1), "conv1x1_bias_act: w must be [Cout, Cin, 1, 1]");
    const int dt = dtype_of(x);
    const int64_t cout = w.size(0), cin = x.size(1), co = (cout + 3) / 4 * 4;
    Tensor w2 = w.reshape({cout, w.size(1)}).to(at::kFloat);
    if (cin != w.size(1) || co != cout) w2 = at::constant_pad_nd(w2, {0, cin - w.size(1), 0, co - cout});
    auto [pt, ct] = tile_for(co, x.size(0) * x.size(2) * x.size(3));
    Tensor wp = pack_matrix(w2, co, cin, 0, dt, ct, x);
    Tensor out = conv_generic(x, wp, padded_bias(bias, cout, ct, x), MAF_OP_CONV1X1, co, x.size(2), x.size(3), 0, 0, (int)act, pt, ct);
    return co == cout ? out : out.narrow(1, 0, cout);
}

Tensor conv1x1_dgrad(const Tensor& dy_in, const Tensor& w) {       // dX = dY . W
    const c10::DeviceGuard device_guard(dy_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    Tensor dy = nhwc(dy_in);
    const int dt = dtype_of(dy);
    const int64_t cout = w.size(0), cin = w.size(1), mult = dt == MAF_F16 ? 8 : 4, kk = (cout + mult - 1) / mult * mult;
    Tensor w2 = w.reshape({cout, cin}).to(at::kFloat);
    if (kk != cout) {                                               // zero-pad the reduction dim to whole 16-byte chunks
        dy = at::constant_pad_nd(dy, {0, 0, 0, 0, 0, kk - cout}).contiguous(at::MemoryFormat::ChannelsLast);
        w2 = at::constant_pad_nd(w2, {0, 0, 0, kk - cout});
    }
    const int64_t ci = (cin + 3) / 4 * 4;
    auto [pt, ct] = tile_for(ci, dy.size(0) * dy.size(2) * dy.size(3));
    if (ci != cin) w2 = at::constant_pad_nd(w2, {0, ci - cin});
    Tensor wp = pack_matrix(w2, kk, ci, 1, dt, ct, dy);
    Tensor out = conv_generic(dy, wp, padded_bias(c10::nullopt, ci, ct, dy), MAF_OP_CONV1X1, ci, dy.size(2), dy.size(3), 0, 0, MAF_ACT_NONE, pt, ct);
    return ci == cin ? out : out.narrow(1, 0, cin);
}

Tensor conv3x3s2_bias_act(const Tensor& x_in, const Tensor& w, const c10::optional<Tensor>& bias, int64_t act) {
    const c10::DeviceGuard device_guard(x_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    Tensor x = pad_channels8(nhwc(x_in));
    TORCH_CHECK(w.dim() == 4 && w.size(2) == 3 && w.size(3) == 3 && w.size(1) == x_in.size(1), "conv3x3s2_bias_act: w must be [Cout, Cin, 3, 3]");
    TORCH_CHECK(w.size(0) % 2 == 0, "conv3x3s2_bias_act: even Cout");
    const int dt = dtype_of(x);
    const int64_t cout = w.size(0), H = x.size(2), W = x.size(3), Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    Tensor wk = x.size(1) != w.size(1) ? at::constant_pad_nd(w, {0, 0, 0, 0, 0, x.size(1) - w.size(1)}) : w;
    auto [pt, ct] = tile_for(cout, x.size(0) * Ho * Wo);
    return conv_generic(x, pack_3x3(wk, false, dt, ct, x), padded_bias(bias, cout, ct, x), MAF_OP_CONV3X3S2, cout, Ho, Wo, H, W, (int)act, pt, ct);
}

Tensor conv3x3s2_dgrad(const Tensor& dy_in, const Tensor& w, int64_t H, int64_t W) {
    const c10::DeviceGuard device_guard(dy_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    Tensor dy = nhwc(dy_in);
    const int dt = dtype_of(dy);
    const int64_t cin = w.size(1), ci = (cin + 7) / 8 * 8;
    Tensor wk = ci != cin ? at::constant_pad_nd(w, {0, 0, 0, 0, 0, ci - cin}) : w;
    auto [pt, ct] = tile_for(ci, dy.size(0) * H * W, true);
    Tensor out = conv_generic(dy, pack_3x3(wk, true, dt, ct, dy), padded_bias(c10::nullopt, ci, ct, dy), MAF_OP_CONV3X3S2_DGRAD, ci, H, W, dy.size(2), dy.size(3), MAF_ACT_NONE, pt, ct);
    return ci == cin ? out : out.narrow(1, 0, cin);
}

Tensor conv_wgrad(const Tensor& x_in, const Tensor& dy_in, int64_t ksize, int64_t stride) {     // fp32 dW [Cout, Cin, k, k]
    const c10::DeviceGuard device_guard(x_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    Tensor x = pad_channels8(nhwc(x_in)), dy = nhwc(dy_in);
    TORCH_CHECK(x.scalar_type() == at::kHalf && dy.scalar_type() == at::kHalf, "conv_wgrad: fp16 activations and gradients (the fp32 parity mode uses the framework's GEMM)");
    const int64_t cout = dy.size(1), co = (cout + 7) / 8 * 8;
    if (co != cout) dy = at::constant_pad_nd(dy, {0, 0, 0, 0, 0, co - cout}).contiguous(at::MemoryFormat::ChannelsLast);
    // the kernel's 3x3 result is tap-major [3][3][Cout][Cin] (csrc/wgrad.hip): permuted to the framework's layout below
    Tensor dw = ksize == 1 ? at::zeros({co, x.size(1), 1, 1}, x.options().dtype(at::kFloat).memory_format(c10::nullopt))
                           : at::zeros({ksize, ksize, co, x.size(1)}, x.options().dtype(at::kFloat).memory_format(c10::nullopt));
    check(maf_conv_wgrad(x.data_ptr(), (int)x.stride(3), dy.data_ptr(), (int)dy.stride(3), (int)x.size(0), (int)dy.size(2), (int)dy.size(3), (int)x.size(2), (int)x.size(3),
                         (int)x.size(1), (int)co, (int)ksize, (int)stride, MAF_F16, dw.data_ptr<float>(), stream_of(x)), "conv_wgrad");
    if (ksize != 1) dw = dw.permute({2, 3, 0, 1});
    return dw.narrow(0, 0, cout).narrow(1, 0, x_in.size(1)).contiguous();
}

Tensor dw_launch(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, int64_t act, int flip) {
    const c10::DeviceGuard device_guard(x.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    const int dt = dtype_of(x);
    const int64_t c = x.size(1), k = w.size(3);
    TORCH_CHECK(w.size(0) == c && w.size(1) == 1 && w.size(2) == k && (k == 3 || k == 5 || k == 7 || k == 9), "dwconv: w must be [C, 1, k, k], k in {3,5,7,9}");
    Tensor wf = w.reshape({c, k * k}).to(at::kFloat).contiguous();
    Tensor wp = at::empty({c * k * k * (dt == MAF_F16 ? 2 : 4)}, x.options().dtype(at::kByte).memory_format(c10::nullopt));
    check(maf_pack_dw(wf.data_ptr<float>(), (int)c, (int)k, flip, dt, wp.data_ptr(), stream_of(x)), "pack_dw");
    Tensor b = at::zeros({c}, x.options().dtype(at::kFloat).memory_format(c10::nullopt));
    if (bias.has_value() && bias->defined()) b.copy_(bias->to(at::kFloat));
    Tensor out = at::empty_like(x, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    maf_op_t op = {};
    op.kind = MAF_OP_DWCONV; op.dtype = dt; op.in_dtype = dt; op.act = (int)act;
    op.B = (int)x.size(0); op.H = (int)x.size(2); op.W = (int)x.size(3); op.Cin = (int)c; op.Cout = (int)c; op.ksize = (int)k;
    fill_src(op, x);
    op.out = out.data_ptr(); op.out_stride = (int)out.stride(3);
    op.w = wp.data_ptr(); op.bias = b.data_ptr<float>();
    check(maf_op_launch(&op, stream_of(x)), "dwconv launch");
    return out;
}

Tensor dwconv_bias_act(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, int64_t act) { return dw_launch(nhwc(x), w, bias, act, 0); }
Tensor dwconv_dgrad(const Tensor& dy, const Tensor& w) { return dw_launch(nhwc(dy), w, c10::nullopt, MAF_ACT_NONE, 1); }      // correlation with the flipped kernel

Tensor dwconv_wgrad(const Tensor& x_in, const Tensor& dy_in, int64_t k) {
    const c10::DeviceGuard device_guard(x_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    Tensor x = nhwc(x_in), dy = nhwc(dy_in);
    const int64_t c = x.size(1), reps = 32;
    Tensor dw = at::zeros({reps, c, k * k}, x.options().dtype(at::kFloat).memory_format(c10::nullopt));
    check(maf_dw_wgrad(x.data_ptr(), (int)x.stride(3), dy.data_ptr(), (int)dy.stride(3), (int)x.size(0), (int)x.size(2), (int)x.size(3), (int)c, (int)k, dtype_of(x),
                       dw.data_ptr<float>(), (int)reps, stream_of(x)), "dw_wgrad");
    return dw.sum(0).reshape({c, 1, k, k});
}

// cls[l] [B, nc, H_l, W_l] (probabilities), reg[l] [B, 4*(reg_max+1), H_l, W_l] (logits), any dtype / layout -> pred fp32 [B, A, 5+nc]
Tensor head_decode(at::TensorList cls, at::TensorList reg, at::ArrayRef<double> strides) {
    TORCH_CHECK(cls.size() > 0, "mafyolo::head_decode: empty level list");
    const c10::DeviceGuard device_guard(cls[0].device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    TORCH_CHECK(cls.size() == 3 && reg.size() == 3 && strides.size() == 3, "head_decode: three detection levels");
    std::vector<Tensor> c(3), r(3);
    maf_op_t op = {};
    op.kind = MAF_OP_DECODE; op.nsrc = 3;
    int64_t A = 0;
    const int64_t B = cls[0].size(0), nc = cls[0].size(1), rs = reg[0].size(1);
    for (int l = 0; l < 3; ++l) {
        TORCH_CHECK(cls[l].is_cuda() && cls[l].size(1) == nc && reg[l].size(1) == rs, "head_decode: level shapes");
        c[l] = cls[l].to(at::kFloat).permute({0, 2, 3, 1}).contiguous();          // [B, H, W, nc] fp32
        r[l] = reg[l].to(at::kFloat).permute({0, 2, 3, 1}).contiguous();
        op.src[l].ptr = c[l].data_ptr(); op.reg[l] = r[l].data_ptr();
        op.lvl_h[l] = (int)cls[l].size(2); op.lvl_w[l] = (int)cls[l].size(3); op.lvl_stride[l] = (float)strides[l];
        A += cls[l].size(2) * cls[l].size(3);
    }
    op.B = (int)B; op.nc = (int)nc; op.reg_stride = (int)rs; op.reg_max = (int)(rs / 4 - 1);
    Tensor pred = at::empty({B, A, 5 + nc}, c[0].options());
    op.out = pred.data_ptr();
    check(maf_op_launch(&op, stream_of(pred)), "decode launch");
    return pred;
}

// -> rows [B, max_det, 6] (x1, y1, x2, y2, conf, cls; rows past counts[b] are unspecified), counts int32 [B]
std::tuple<Tensor, Tensor> decode_nms(const Tensor& pred_in, double conf, double iou, bool agnostic, bool multi_label, int64_t max_det, at::OptionalIntArrayRef classes) {
    const c10::DeviceGuard device_guard(pred_in.device());      // ops get no automatic device guard: kernels, allocations and the stream follow the tensor's device
    TORCH_CHECK(conf >= 0 && conf <= 1, "conf_thresh must be in 0.0 to 1.0, however ", conf, " is provided.");
    TORCH_CHECK(iou >= 0 && iou <= 1, "iou_thres must be in 0.0 to 1.0, however ", iou, " is provided.");
    TORCH_CHECK(pred_in.is_cuda() && pred_in.dim() == 3, "decode_nms: prediction [B, N, 5+nc] on the HIP device");
    Tensor pred = pred_in.to(at::kFloat).contiguous();
    const int64_t B = pred.size(0), N = pred.size(1), nc = pred.size(2) - 5;
    Tensor ws = at::empty({maf_nms_workspace_bytes((int)B, (int)N, (int)nc)}, pred.options().dtype(at::kByte));
    Tensor rows = at::empty({B, max_det, 6}, pred.options()), idx = at::empty({B, max_det}, pred.options().dtype(at::kLong)), cnt = at::empty({B}, pred.options().dtype(at::kInt));
    Tensor cls_t;
    int ncls = 0;
    if (classes.has_value()) {
        std::vector<int32_t> v(classes->begin(), classes->end());
        ncls = (int)v.size();
        if (ncls) cls_t = at::tensor(v, at::kInt).to(pred.device());
    }
    check(maf_nms(pred.data_ptr<float>(), (int)B, (int)N, (int)nc, conf, iou, ncls ? cls_t.data_ptr<int32_t>() : nullptr, ncls, agnostic ? 1 : 0, multi_label ? 1 : 0,
                  (int)max_det, ws.data_ptr(), ws.numel(), rows.data_ptr<float>(), idx.data_ptr<int64_t>(), cnt.data_ptr<int32_t>(), stream_of(pred)), "nms");
    return {rows, cnt};
}

// One launch of `kind` writing channels [coff, coff + cout) of an existing NHWC tensor `out` (the concat buffers of mprep / sppf)
void conv_into(const Tensor& x, int src_mode, const Tensor& wp, const Tensor& bias, int kind, int64_t cout, int64_t Hin, int64_t Win, int act, int pt, int ct,
               Tensor& out, int64_t coff) {
    const int dt = dtype_of(x);
    maf_op_t op = {};
    op.kind = kind; op.dtype = dt; op.in_dtype = dt; op.act = act;
    op.B = (int)out.size(0); op.H = (int)out.size(2); op.W = (int)out.size(3); op.Hin = (int)Hin; op.Win = (int)Win; op.Cin = (int)x.size(1); op.Cout = (int)cout;
    fill_src(op, x, src_mode);
    op.out = out.data_ptr(); op.out_stride = (int)out.stride(3); op.out_coff = (int)coff;
    op.tile_p = pt; op.tile_c = ct;
    op.w = wp.data_ptr(); op.bias = bias.data_ptr<float>();
    check(maf_op_launch(&op, stream_of(x)), "conv launch");
}

Tensor pack_1x1(const Tensor& w, int64_t cin_x, int dt, int ct, const Tensor& like) {
    const int64_t cout = w.size(0);
    Tensor w2 = w.reshape({cout, w.size(1)}).to(at::kFloat);
    if (cin_x != w.size(1)) w2 = at::constant_pad_nd(w2, {0, cin_x - w.size(1)});
    return pack_matrix(w2, cout, cin_x, 0, dt, ct, like);
}

// MPRep in deploy form (yolov6/layers/common.py:776-792): cat([Conv1x1+SiLU(MaxPool2x2(x)), RepVGG3x3s2+ReLU(x)], 1) — two launches, both writing
// their slice of ONE output tensor; the 2x2 max-pool is folded into the operand load of the 1x1 (MAF_SRC_POOL2), no pooled tensor, no cat.
Tensor mprep(const Tensor& x_in, const Tensor& w1, const c10::optional<Tensor>& b1, const Tensor& w3, const c10::optional<Tensor>& b3) {
    const c10::DeviceGuard device_guard(x_in.device());
    Tensor x = nhwc(x_in);
    TORCH_CHECK(x.size(1) % 8 == 0 && x.size(2) % 2 == 0 && x.size(3) % 2 == 0, "mprep: channels in whole 16-byte groups, even sides");
    TORCH_CHECK(w1.dim() == 4 && w1.size(2) == 1 && w1.size(1) == x.size(1) && w3.dim() == 4 && w3.size(2) == 3 && w3.size(1) == x.size(1), "mprep: w1 [c, Cin, 1, 1], w3 [c, Cin, 3, 3]");
    const int dt = dtype_of(x);
    const int64_t c1 = w1.size(0), c3 = w3.size(0), Ho = x.size(2) / 2, Wo = x.size(3) / 2, M = x.size(0) * Ho * Wo;
    TORCH_CHECK(c1 % 8 == 0 && c3 % 8 == 0, "mprep: branch widths in whole 16-byte groups");
    Tensor out = at::empty({x.size(0), c1 + c3, Ho, Wo}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    auto [pt1, ct1] = tile_for(c1, M);
    conv_into(x, MAF_SRC_POOL2, pack_1x1(w1, x.size(1), dt, ct1, x), padded_bias(b1, c1, ct1, x), MAF_OP_CONV1X1, c1, 0, 0, MAF_ACT_SILU, pt1, ct1, out, 0);
    auto [pt3, ct3] = tile_for(c3, M);
    conv_into(x, MAF_SRC_DIRECT, pack_3x3(w3, false, dt, ct3, x), padded_bias(b3, c3, ct3, x), MAF_OP_CONV3X3S2, c3, x.size(2), x.size(3), MAF_ACT_RELU, pt3, ct3, out, c1);
    return out;
}

// SPPF in deploy form (common.py:114-129): cv2(cat([y, m(y), m(m(y)), m(m(m(y)))])) with y = cv1(x), m = MaxPool 5x5 s1 p2 — cv1 writes slice 0 of the
// concat buffer, ONE pooling launch (three cascaded separable 5-max passes in LDS) writes the other three slices, cv2 reads the buffer.
Tensor sppf(const Tensor& x_in, const Tensor& w1, const c10::optional<Tensor>& b1, const Tensor& w2, const c10::optional<Tensor>& b2) {
    const c10::DeviceGuard device_guard(x_in.device());
    Tensor x = nhwc(x_in);
    const int dt = dtype_of(x);
    const int64_t c_ = w1.size(0), cout = w2.size(0), H = x.size(2), W = x.size(3), M = x.size(0) * H * W;
    TORCH_CHECK(w1.size(1) == x.size(1) && w2.size(1) == 4 * c_ && c_ % 8 == 0 && x.size(1) % 8 == 0 && cout % 2 == 0, "sppf: cv1 [c_, Cin, 1, 1], cv2 [Cout, 4 c_, 1, 1]");
    Tensor cat = at::empty({x.size(0), 4 * c_, H, W}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    auto [pt1, ct1] = tile_for(c_, M);
    conv_into(x, MAF_SRC_DIRECT, pack_1x1(w1, x.size(1), dt, ct1, x), padded_bias(b1, c_, ct1, x), MAF_OP_CONV1X1, c_, 0, 0, MAF_ACT_SILU, pt1, ct1, cat, 0);
    {
        maf_op_t op = {};
        op.kind = MAF_OP_SPPF_POOL; op.dtype = dt; op.in_dtype = dt;
        op.B = (int)x.size(0); op.H = (int)H; op.W = (int)W; op.Cin = (int)c_; op.Cout = (int)(3 * c_);
        op.nsrc = 1;
        op.src[0].ptr = cat.data_ptr(); op.src[0].C = (int)c_; op.src[0].stride = (int)cat.stride(3); op.src[0].coff = 0; op.src[0].mode = MAF_SRC_DIRECT;
        op.out = cat.data_ptr(); op.out_stride = (int)cat.stride(3); op.out_coff = (int)c_;
        check(maf_op_launch(&op, stream_of(x)), "sppf pool launch");
    }
    auto [pt2, ct2] = tile_for(cout, M);
    return conv_generic(cat, pack_1x1(w2, 4 * c_, dt, ct2, x), padded_bias(b2, cout, ct2, x), MAF_OP_CONV1X1, cout, H, W, 0, 0, MAF_ACT_SILU, pt2, ct2);
}

// BatchNorm2d(train) + activation (Conv.forward = act(bn(conv(x))), common.py:44-47) on csrc/bn_act.hip: batch statistics, running-stat update
// (torch's momentum rule, unbiased variance) -> (y, save_mean, save_rstd, new_running_mean, new_running_var).  The op is FUNCTIONAL (autograd formulas
// are registered for functional ops only): the updated running statistics come back as new tensors (copies of the inputs when given, else empty);
// torch_ops.bn_act_ copies them over the module's buffers.
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> bn_act(const Tensor& x_in, const Tensor& gamma, const Tensor& beta, const c10::optional<Tensor>& running_mean,
                                           const c10::optional<Tensor>& running_var, double eps, double momentum, int64_t act) {
    const c10::DeviceGuard device_guard(x_in.device());
    Tensor x = nhwc(x_in);
    const int dt = dtype_of(x);
    const int64_t c = x.size(1), M = x.size(0) * x.size(2) * x.size(3);
    TORCH_CHECK(c % (dt == MAF_F16 ? 8 : 4) == 0, "bn_act: channels in whole 16-byte groups");
    Tensor y = at::empty_like(x, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    Tensor mean = at::empty({c}, x.options().dtype(at::kFloat).memory_format(c10::nullopt)), rstd = at::empty_like(mean);
    const int R = 16;
    Tensor part = at::zeros({2 * R * 2 * ((c + 255) / 256 * 256)}, mean.options());
    Tensor g = gamma.to(at::kFloat).contiguous(), b = beta.to(at::kFloat).contiguous();
    Tensor nrm = running_mean.has_value() && running_mean->defined() ? running_mean->to(at::kFloat).clone() : at::empty({0}, mean.options());
    Tensor nrv = running_var.has_value() && running_var->defined() ? running_var->to(at::kFloat).clone() : at::empty({0}, mean.options());
    float* rm = nrm.numel() ? nrm.data_ptr<float>() : nullptr;
    float* rv = nrv.numel() ? nrv.data_ptr<float>() : nullptr;
    check(maf_bn_forward(x.data_ptr(), (int)x.stride(3), (int)M, (int)c, dt, g.data_ptr<float>(), b.data_ptr<float>(), (float)eps, (float)momentum, rm, rv, nullptr,
                         (int)act, y.data_ptr(), (int)y.stride(3), mean.data_ptr<float>(), rstd.data_ptr<float>(), part.data_ptr<float>(), R, 0, nullptr, 0, stream_of(x)),
          "bn_forward");
    return {y, mean, rstd, nrm, nrv};
}

// -> (dx, dgamma, dbeta) from dz = the gradient with respect to the activation output
std::tuple<Tensor, Tensor, Tensor> bn_act_backward(const Tensor& x_in, const Tensor& dz_in, const Tensor& gamma, const Tensor& beta, const Tensor& save_mean,
                                                    const Tensor& save_rstd, int64_t act) {
    const c10::DeviceGuard device_guard(x_in.device());
    Tensor x = nhwc(x_in), dz = nhwc(dz_in.to(x_in.scalar_type()));
    const int dt = dtype_of(x);
    const int64_t c = x.size(1), M = x.size(0) * x.size(2) * x.size(3);
    Tensor dx = at::empty_like(x, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    Tensor dg = at::empty({c}, x.options().dtype(at::kFloat).memory_format(c10::nullopt)), db = at::empty_like(dg);
    const int R = 16;
    Tensor part = at::zeros({2 * R * 2 * ((c + 255) / 256 * 256)}, dg.options());
    Tensor g = gamma.to(at::kFloat).contiguous(), b = beta.to(at::kFloat).contiguous();
    check(maf_bn_backward(x.data_ptr(), (int)x.stride(3), dz.data_ptr(), (int)dz.stride(3), (int)M, (int)c, dt, g.data_ptr<float>(), b.data_ptr<float>(),
                          save_mean.data_ptr<float>(), save_rstd.data_ptr<float>(), (int)act, dx.data_ptr(), (int)dx.stride(3), dg.data_ptr<float>(), db.data_ptr<float>(),
                          part.data_ptr<float>(), R, 0, nullptr, 0, nullptr, 0, stream_of(x)), "bn_backward");
    return {dx, dg, db};
}

}  // namespace

TORCH_LIBRARY(mafyolo, m) {
    m.def("mprep(Tensor x, Tensor w1, Tensor? b1, Tensor w3, Tensor? b3) -> Tensor");
    m.def("sppf(Tensor x, Tensor w1, Tensor? b1, Tensor w2, Tensor? b2) -> Tensor");
    m.def("bn_act(Tensor x, Tensor gamma, Tensor beta, Tensor? running_mean, Tensor? running_var, float eps, float momentum, int act) -> (Tensor, Tensor, Tensor, Tensor, Tensor)");
    m.def("bn_act_backward(Tensor x, Tensor dz, Tensor gamma, Tensor beta, Tensor save_mean, Tensor save_rstd, int act) -> (Tensor, Tensor, Tensor)");
    m.def("conv1x1_bias_act(Tensor x, Tensor w, Tensor? bias, int act) -> Tensor");
    m.def("conv3x3s2_bias_act(Tensor x, Tensor w, Tensor? bias, int act) -> Tensor");
    m.def("dwconv_bias_act(Tensor x, Tensor w, Tensor? bias, int act) -> Tensor");
    m.def("conv1x1_dgrad(Tensor dy, Tensor w) -> Tensor");
    m.def("conv3x3s2_dgrad(Tensor dy, Tensor w, int H, int W) -> Tensor");
    m.def("conv_wgrad(Tensor x, Tensor dy, int ksize, int stride) -> Tensor");
    m.def("dwconv_dgrad(Tensor dy, Tensor w) -> Tensor");
    m.def("dwconv_wgrad(Tensor x, Tensor dy, int k) -> Tensor");
    m.def("head_decode(Tensor[] cls, Tensor[] reg, float[] strides) -> Tensor");
    m.def("decode_nms(Tensor pred, float conf_thres, float iou_thres, bool agnostic, bool multi_label, int max_det, int[]? classes) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(mafyolo, CUDA, m) {       // the HIP device is the "CUDA" dispatch key of PyTorch-ROCm
    m.impl("mprep", &mprep);
    m.impl("sppf", &sppf);
    m.impl("bn_act", &bn_act);
    m.impl("bn_act_backward", &bn_act_backward);
    m.impl("conv1x1_bias_act", &conv1x1_bias_act);
    m.impl("conv3x3s2_bias_act", &conv3x3s2_bias_act);
    m.impl("dwconv_bias_act", &dwconv_bias_act);
    m.impl("conv1x1_dgrad", &conv1x1_dgrad);
    m.impl("conv3x3s2_dgrad", &conv3x3s2_dgrad);
    m.impl("conv_wgrad", &conv_wgrad);
    m.impl("dwconv_dgrad", &dwconv_dgrad);
    m.impl("dwconv_wgrad", &dwconv_wgrad);
    m.impl("head_decode", &head_decode);
    m.impl("decode_nms", &decode_nms);
}
