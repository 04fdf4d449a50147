// Post-NMS tail on the device: for every surviving detection of a batch, rescale the box from the
// network-input frame to the original image, clamp, convert to COCO's (x, y, w, h) and look the category id up — one
// launch and one packed array instead of the reference's Python loop with .tolist() / .item() per box.
//
// Replaces Evaler.scale_coords (yolov6/core/evaler.py:382-409, the ratio_pad branch), Evaler.box_convert (:374-381) and
// the tensor part of Evaler.convert_to_coco_format (:411-420).  Arithmetic is fp32 in the reference's operation order
// (subtract pad, IEEE divide by the gain, clamp, centre/extent, centre - extent/2), so the rows are bit-identical; the decimal
// rounding of :425-426 stays on the host (maf-yolo_amd/post.py), where it is one vectorised numpy call.
#include "maf_common.h"

namespace {

struct PostArgs {
    const float* rows; const int* count; const float* img; const int* ids;
    float* out; int* total;
    int B, max_det, n_ids;
};

__global__ __launch_bounds__(256) void coco_rows_kernel(const PostArgs a) {
    const int b = blockIdx.x;
    int base = 0;
    for (int i = 0; i < b; ++i) base += min(a.count[i], a.max_det);
    const int n = min(a.count[b], a.max_det);
    if (b == a.B - 1 && threadIdx.x == 0) *a.total = base + n;
    const float h0 = a.img[b * 6 + 0], w0 = a.img[b * 6 + 1], gx = a.img[b * 6 + 2], gy = a.img[b * 6 + 3];
    const float pw = a.img[b * 6 + 4], ph = a.img[b * 6 + 5];
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const float* r = a.rows + ((size_t)b * a.max_det + k) * 6;
        float x1 = (r[0] - pw) / gx, y1 = (r[1] - ph) / gy, x2 = (r[2] - pw) / gx, y2 = (r[3] - ph) / gy;
        x1 = fminf(fmaxf(x1, 0.f), w0); x2 = fminf(fmaxf(x2, 0.f), w0);
        y1 = fminf(fmaxf(y1, 0.f), h0); y2 = fminf(fmaxf(y2, 0.f), h0);
        const float cx = (x1 + x2) / 2.f, cy = (y1 + y2) / 2.f, w = x2 - x1, h = y2 - y1;
        const int cls = (int)r[5];
        float* o = a.out + (size_t)(base + k) * 7;
        o[0] = (float)b;
        o[1] = (float)((cls >= 0 && cls < a.n_ids) ? a.ids[cls] : cls);
        o[2] = cx - w / 2.f; o[3] = cy - h / 2.f; o[4] = w; o[5] = h; o[6] = r[4];
    }
}

}  // namespace

extern "C" int maf_coco_rows(const float* rows, const int32_t* count, int32_t B, int32_t max_det, const float* img_params,
                             const int32_t* ids, int32_t n_ids, float* out, int32_t* out_total, maf_stream_t stream) {
    MAF_REQUIRE(rows && count && img_params && out && out_total, "coco_rows: null pointer");
    MAF_REQUIRE(B > 0 && max_det > 0 && (n_ids == 0 || ids), "coco_rows: bad shape");
    PostArgs a;
    a.rows = rows; a.count = count; a.img = img_params; a.ids = ids; a.out = out; a.total = out_total;
    a.B = B; a.max_det = max_det; a.n_ids = n_ids;
    hipLaunchKernelGGL(coco_rows_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return maf_check_hip(hipGetLastError(), "coco_rows launch");
}
