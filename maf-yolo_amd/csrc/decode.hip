// Detect head decode (eval branch): DFL softmax expectation, anchor points, dist2bbox('xywh'),
// x stride, objectness = 1, class probabilities -> pred fp32 [B, A, 5+nc].
//
// Replaces Detect_yaml.forward eval branch (yolov6/models/yolo.py:355-396) =
// generate_anchors(is_eval=True) (yolov6/assigners/anchor_generator.py:11-25) + F.softmax +
// proj_conv (yolo.py:377-378, weights linspace(0,16,17) :327-330) + dist2bbox (yolov6/utils/general.py:29-40)
// + the three torch.cat calls (yolo.py:383-396).
//
// One workgroup decodes 64 consecutive anchors of one image and level: phase 1 = 256 lanes, one
// (anchor, side) each, 17-bin softmax expectation in fp32 from the NHWC reg logits; phase 2 = the
// 64 x (5+nc) output floats are one contiguous run of the prediction tensor, written (and the class
// probabilities read) fully coalesced.
#include "maf_common.h"

namespace {

struct DecArgs {
    const float* cls[3];
    const float* reg[3];
    int lvl_h[3], lvl_w[3], lvl_off[4], lvl_blk[4];
    float lvl_stride[3];
    float* out;
    int B, A, nc, reg_stride, reg_max, cls_stride[3];
};

template <int NO_CT>   // 5+nc known at compile time (85) => the row/col split is a multiply-shift, not an integer division
__global__ __launch_bounds__(256) void decode_kernel(const DecArgs a) {
    __shared__ float box[64][4];
    __shared__ __attribute__((aligned(16))) float regs[64 * 68 + 4];   // the block's reg logits, staged with coalesced loads
    const int b = blockIdx.y;
    int blk = blockIdx.x;
    const int l = (blk >= a.lvl_blk[1]) + (blk >= a.lvl_blk[2]);
    blk -= a.lvl_blk[l];
    const int L = a.lvl_h[l] * a.lvl_w[l];
    if ((l == 0 ? a.cls[0] : l == 1 ? a.cls[1] : a.cls[2]) == nullptr) return;   // this level's rows were written by MAF_OP_HEADTAIL
    const int a0 = blk * 64;                       // first anchor (within level)
    const int nA = min(64, L - a0);
    const float* reg = (l == 0 ? a.reg[0] : l == 1 ? a.reg[1] : a.reg[2]) + ((size_t)b * L + a0) * a.reg_stride;
    const int cstr = l == 0 ? a.cls_stride[0] : l == 1 ? a.cls_stride[1] : a.cls_stride[2];      // >= nc: the pred conv pads its rows to 4 channels
    const float* cls = (l == 0 ? a.cls[0] : l == 1 ? a.cls[1] : a.cls[2]) + ((size_t)b * L + a0) * cstr;
    const int W = l == 0 ? a.lvl_w[0] : l == 1 ? a.lvl_w[1] : a.lvl_w[2];
    const float stride = l == 0 ? a.lvl_stride[0] : l == 1 ? a.lvl_stride[1] : a.lvl_stride[2];
    const int tid = threadIdx.x;
    const bool staged = a.reg_stride == 68 && a.reg_max == 16;              // the MAF-YOLO head: 64 x 68 contiguous floats
    if (staged) {
        const int n4 = nA * 17;                                             // float4 count (68 floats = 17 float4 per anchor)
        const f32x4_t* src = reinterpret_cast<const f32x4_t*>(reg);
        for (int i = tid; i < n4; i += 256) *reinterpret_cast<f32x4_t*>(&regs[4 * i]) = src[i];
        __syncthreads();
    }
    {
        const int ai = tid >> 2, side = tid & 3;
        if (ai < nA) {
            const int nb = a.reg_max + 1;
            const float* r = staged ? &regs[ai * 68 + side * 17] : reg + (size_t)ai * a.reg_stride + side * nb;   // channel = side*17 + bin (yolo.py:376)
            float mx = -INFINITY;
            for (int i = 0; i < nb; ++i) mx = fmaxf(mx, r[i]);
            float se = 0.f, sw = 0.f;
            for (int i = 0; i < nb; ++i) {
                const float e = __expf(r[i] - mx);
                se += e;
                sw += e * (float)i;
            }
            box[ai][side] = sw / se;                                        // expected ltrb distance in grid units
        }
    }
    __syncthreads();
    const int no = NO_CT > 0 ? NO_CT : 5 + a.nc;
    float* out = a.out + ((size_t)b * a.A + a.lvl_off[l] + a0) * no;
    for (int e = tid; e < nA * no; e += 256) {
        const int ai = e / no, col = e - ai * no;
        float v;
        if (col < 4) {
            const int idx = a0 + ai;
            const float ax = (float)(idx % W) + 0.5f, ay = (float)(idx / W) + 0.5f;
            const float lft = box[ai][0], top = box[ai][1], rgt = box[ai][2], bot = box[ai][3];
            const float x1 = ax - lft, y1 = ay - top, x2 = ax + rgt, y2 = ay + bot;
            v = col == 0 ? (x1 + x2) * 0.5f : col == 1 ? (y1 + y2) * 0.5f : col == 2 ? x2 - x1 : y2 - y1;
            v *= stride;
        } else if (col == 4) {
            v = 1.0f;
        } else {
            v = cls[(size_t)ai * cstr + (col - 5)];
        }
        out[e] = v;
    }
}

}  // namespace

int maf_launch_decode(const maf_op_t* op, hipStream_t s) {
    MAF_REQUIRE(op->nsrc == 3, "decode: three levels");
    MAF_REQUIRE(op->nc > 0 && op->reg_max > 0 && op->reg_stride >= 4 * (op->reg_max + 1), "decode: bad nc/reg_max/reg_stride");
    MAF_REQUIRE(op->out, "decode: null out");
    DecArgs a;
    int off = 0, blk = 0;
    for (int l = 0; l < 3; ++l) {
        MAF_REQUIRE((op->src[l].ptr != nullptr) == (op->reg[l] != nullptr) && op->lvl_h[l] > 0 && op->lvl_w[l] > 0, "decode: cls and reg of a level come together (both null: level decoded elsewhere)");
        a.cls[l] = static_cast<const float*>(op->src[l].ptr);
        a.reg[l] = static_cast<const float*>(op->reg[l]);
        a.lvl_h[l] = op->lvl_h[l]; a.lvl_w[l] = op->lvl_w[l]; a.lvl_stride[l] = op->lvl_stride[l];
        a.cls_stride[l] = op->src[l].stride > 0 ? op->src[l].stride : op->nc;
        MAF_REQUIRE(a.cls_stride[l] >= op->nc, "decode: src[l].stride (row stride of the class scores; 0 = nc) is smaller than nc");
        a.lvl_off[l] = off; a.lvl_blk[l] = blk;
        off += op->lvl_h[l] * op->lvl_w[l];
        blk += maf_cdiv(op->lvl_h[l] * op->lvl_w[l], 64);
    }
    a.lvl_off[3] = off; a.lvl_blk[3] = blk;
    a.out = static_cast<float*>(op->out);
    a.B = op->B; a.A = off; a.nc = op->nc; a.reg_stride = op->reg_stride; a.reg_max = op->reg_max;
    if (op->nc == 80) hipLaunchKernelGGL(decode_kernel<85>, dim3(blk, op->B), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(decode_kernel<0>, dim3(blk, op->B), dim3(256), 0, s, a);
    return maf_check_hip(hipGetLastError(), "decode launch");
}
