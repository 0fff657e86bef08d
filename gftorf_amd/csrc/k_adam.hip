// k_adam.hip -- fused Adam step (gfx950): one pass, 16-byte accesses, HBM-bound (28 B per element).
// Arithmetic of torch/optim/adam.py `_single_tensor_adam` (no amsgrad / maximize), which the
// reference's optimizer follows (scene/gaussian_model.py:274, train.py:470).
#include "gft_internal.h"
#include "gftorf_optim.h"

#include <cmath>

namespace {

#define ADAM_BLOCK 256
#define ADAM_ITEMS 2      // 16-byte groups per thread of the dense multi-tensor kernel: 8 loads of 16 B in flight per thread

struct AdamArgs {
    int64_t n;
    float* p; const float* g; float* m; float* v;
    float one_m_beta1, beta2, one_m_beta2, step_size, bias2_sqrt, eps, weight_decay;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a)
{
#pragma clang fp contract(off)
    if (a.weight_decay != 0.f) g = g + a.weight_decay * p;
    m = m + a.one_m_beta1 * (g - m);                          // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (a.one_m_beta2 * g) * g;                // mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
    p = p + (-a.step_size) * (m / denom);                     // addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(ADAM_BLOCK) void k_adam_step(AdamArgs a)
{
    const int64_t n4 = a.n >> 2;
    const int64_t i = (int64_t)blockIdx.x * ADAM_BLOCK + threadIdx.x;
    if (i < n4) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        const float4 g = reinterpret_cast<const float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        adam_one(p.x, g.x, m.x, v.x, a);
        adam_one(p.y, g.y, m.y, v.y, a);
        adam_one(p.z, g.z, m.z, v.z, a);
        adam_one(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
    }
    // tail (n not a multiple of 4): the first workgroup's first lanes
    const int64_t tail = a.n & 3;
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < tail) {
        const int64_t k = (n4 << 2) + threadIdx.x;
        float p = a.p[k], m = a.m[k], v = a.v[k];
        adam_one(p, a.g[k], m, v, a);
        a.p[k] = p; a.m[k] = m; a.v[k] = v;
    }
}

// Several tensors in ONE launch (own learning rate and step count each; betas, eps, weight decay shared): the
// reference's optimizers hold 13 one-tensor groups of per-Gaussian parameters and the 36 tensors of the deformation
// network; at 100 k Gaussians a launch per tensor is bound by launch latency, not by HBM.  Workgroup b serves the
// tensor whose block range holds b (table in the kernel arguments).
struct AdamMultiArgs {
    int count;
    float one_m_beta1, beta2, one_m_beta2, eps, weight_decay;
    struct T { float* p; const float* g; float* m; float* v; int64_t n; float step_size, bias2_sqrt; uint32_t first_block; uint32_t pad; } t[GFT_ADAM_MAX_TENSORS];
};

// Learning rates and step counts on the device (gft_adam_step_multi_dev): one workgroup, thread c advances step[c] and leaves
// the two derived factors of tensor c where the update kernel behind it reads them.
struct AdamTickArgs {
    int count;
    double beta1, beta2;
    float* factors;
    const double* lr[GFT_ADAM_MAX_TENSORS];
    float* step[GFT_ADAM_MAX_TENSORS];
};

__global__ __launch_bounds__(64) void k_adam_tick(AdamTickArgs a)
{
    const int c = threadIdx.x;
    if (c >= a.count) return;
    const float t = *a.step[c] + 1.0f;
    *a.step[c] = t;
    const double lr = *a.lr[c];
    // torch/optim/adam.py: step_size = lr / (1 - beta1 ** step); bias_correction2_sqrt = (1 - beta2 ** step) ** 0.5
    a.factors[2 * c] = (float)(lr / (1.0 - pow(a.beta1, (double)t)));
    a.factors[2 * c + 1] = (float)sqrt(1.0 - pow(a.beta2, (double)t));
}

template <bool DEV>
__global__ __launch_bounds__(ADAM_BLOCK) void k_adam_multi(AdamMultiArgs a, const float* __restrict__ factors)
{
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < a.count; q++)
        if (blockIdx.x >= a.t[q].first_block) k = q;
    AdamArgs s;
    s.n = a.t[k].n; s.p = a.t[k].p; s.g = a.t[k].g; s.m = a.t[k].m; s.v = a.t[k].v;
    s.one_m_beta1 = a.one_m_beta1; s.beta2 = a.beta2; s.one_m_beta2 = a.one_m_beta2; s.step_size = a.t[k].step_size;
    s.bias2_sqrt = a.t[k].bias2_sqrt; s.eps = a.eps; s.weight_decay = a.weight_decay;
    if (DEV) {          // (t[k].pad: the tensor's index in the caller's table = its slot in `factors`)
        s.step_size = factors[2 * a.t[k].pad];
        s.bias2_sqrt = factors[2 * a.t[k].pad + 1];
    }
    const uint32_t blk = blockIdx.x - a.t[k].first_block;
    const int64_t n4 = s.n >> 2;
    // ADAM_ITEMS 16-byte groups per thread, all their loads issued before the arithmetic (index clamped, the stores
    // predicated): one group per thread kept too few bytes in flight for the HBM rate (3.7 TB/s)
    float4 p[ADAM_ITEMS], g[ADAM_ITEMS], m[ADAM_ITEMS], v[ADAM_ITEMS];
    int64_t idx[ADAM_ITEMS];
#pragma unroll
    for (int u = 0; u < ADAM_ITEMS; u++) {
        idx[u] = ((int64_t)blk * ADAM_ITEMS + u) * ADAM_BLOCK + threadIdx.x;
        const int64_t i = n4 > 0 ? (idx[u] < n4 ? idx[u] : n4 - 1) : 0;
        if (n4 > 0) {
            p[u] = reinterpret_cast<float4*>(s.p)[i];
            g[u] = reinterpret_cast<const float4*>(s.g)[i];
            m[u] = reinterpret_cast<float4*>(s.m)[i];
            v[u] = reinterpret_cast<float4*>(s.v)[i];
        }
    }
#pragma unroll
    for (int u = 0; u < ADAM_ITEMS; u++) {
        if (idx[u] < n4) {
            adam_one(p[u].x, g[u].x, m[u].x, v[u].x, s);
            adam_one(p[u].y, g[u].y, m[u].y, v[u].y, s);
            adam_one(p[u].z, g[u].z, m[u].z, v[u].z, s);
            adam_one(p[u].w, g[u].w, m[u].w, v[u].w, s);
            reinterpret_cast<float4*>(s.p)[idx[u]] = p[u];
            reinterpret_cast<float4*>(s.m)[idx[u]] = m[u];
            reinterpret_cast<float4*>(s.v)[idx[u]] = v[u];
        }
    }
    const int64_t tail = s.n & 3;
    if (blk == 0 && (int64_t)threadIdx.x < tail) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        float pe = s.p[e], me = s.m[e], ve = s.v[e];
        adam_one(pe, s.g[e], me, ve, s);
        s.p[e] = pe; s.m[e] = me; s.v[e] = ve;
    }
}

// The same launch for tensors whose rows are Gaussians, restricted to the rows of a mask (opt-in: FusedAdam.step(visibility=...);
// the reference's optimizer is dense): a 16-byte group none of whose elements lies in a masked row is neither read
// nor written -- with 14 % of the Gaussians on screen the step moves 14 % of the bytes.
struct AdamRowsArgs {
    AdamMultiArgs m;
    const uint8_t* row_mask;
    uint32_t row_floats[GFT_ADAM_MAX_TENSORS];
};

__global__ __launch_bounds__(ADAM_BLOCK) void k_adam_rows(AdamRowsArgs a)
{
    int k = 0;
#pragma unroll 1
    for (int q = 1; q < a.m.count; q++)
        if (blockIdx.x >= a.m.t[q].first_block) k = q;
    AdamArgs s;
    s.n = a.m.t[k].n; s.p = a.m.t[k].p; s.g = a.m.t[k].g; s.m = a.m.t[k].m; s.v = a.m.t[k].v;
    s.one_m_beta1 = a.m.one_m_beta1; s.beta2 = a.m.beta2; s.one_m_beta2 = a.m.one_m_beta2; s.step_size = a.m.t[k].step_size;
    s.bias2_sqrt = a.m.t[k].bias2_sqrt; s.eps = a.m.eps; s.weight_decay = a.m.weight_decay;
    const uint32_t rf = a.row_floats[k];
    const uint32_t blk = blockIdx.x - a.m.t[k].first_block;
    const int64_t n4 = s.n >> 2;
    const int64_t i = (int64_t)blk * ADAM_BLOCK + threadIdx.x;
    if (i < n4) {
        const uint64_t e0 = (uint64_t)i << 2;
        uint64_t row = e0 / rf;
        uint32_t rem = (uint32_t)(e0 - row * rf);
        bool on[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            on[j] = a.row_mask[row] != 0;
            if (++rem == rf) { rem = 0; row++; }
        }
        if (on[0] | on[1] | on[2] | on[3]) {
            float4 p = reinterpret_cast<float4*>(s.p)[i];
            const float4 g = reinterpret_cast<const float4*>(s.g)[i];
            float4 m = reinterpret_cast<float4*>(s.m)[i];
            float4 v = reinterpret_cast<float4*>(s.v)[i];
            float4 p1 = p, m1 = m, v1 = v;
            adam_one(p1.x, g.x, m1.x, v1.x, s);
            adam_one(p1.y, g.y, m1.y, v1.y, s);
            adam_one(p1.z, g.z, m1.z, v1.z, s);
            adam_one(p1.w, g.w, m1.w, v1.w, s);
            if (on[0]) { p.x = p1.x; m.x = m1.x; v.x = v1.x; }
            if (on[1]) { p.y = p1.y; m.y = m1.y; v.y = v1.y; }
            if (on[2]) { p.z = p1.z; m.z = m1.z; v.z = v1.z; }
            if (on[3]) { p.w = p1.w; m.w = m1.w; v.w = v1.w; }
            reinterpret_cast<float4*>(s.p)[i] = p;
            reinterpret_cast<float4*>(s.m)[i] = m;
            reinterpret_cast<float4*>(s.v)[i] = v;
        }
    }
    const int64_t tail = s.n & 3;
    if (blk == 0 && (int64_t)threadIdx.x < tail) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        if (a.row_mask[(uint64_t)e / rf]) {
            float p = s.p[e], m = s.m[e], v = s.v[e];
            adam_one(p, s.g[e], m, v, s);
            s.p[e] = p; s.m[e] = m; s.v[e] = v;
        }
    }
}

}  // namespace

// fills the per-tensor table of a launch from tensors[c0 ...]; returns the number of entries (< 0: error)
static int adam_table(AdamMultiArgs& a, const gft_adam_tensor* tensors, int32_t c0, int32_t count, double beta1, double beta2,
                      double eps, double weight_decay, uint64_t* blocks_out, int64_t rows, uint32_t* row_floats, const char* who,
                      bool dev = false)
{
    a.one_m_beta1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.one_m_beta2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    a.weight_decay = (float)weight_decay;
    int k = 0;
    uint64_t blocks = 0;
    for (int32_t c = c0; c < count && c < c0 + GFT_ADAM_MAX_TENSORS; c++) {
        const gft_adam_tensor& t = tensors[c];
        if (t.n < 0) { gft_fail("%s: tensor %d has n < 0", who, c); return -1; }
        if (t.n == 0) continue;
        if (!dev && t.step < 1) { gft_fail("%s: tensor %d: step must be >= 1", who, c); return -1; }
        if (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq) { gft_fail("%s: tensor %d has a NULL pointer", who, c); return -1; }
        if ((((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) != 0) {
            gft_fail("%s: pointers of tensor %d are not 16-byte aligned", who, c);
            return -1;
        }
        if (row_floats) {
            if (t.n % rows != 0 || t.n / rows > 0xffffffffll) { gft_fail("%s: tensor %d does not have %lld rows", who, c, (long long)rows); return -1; }
            row_floats[k] = (uint32_t)(t.n / rows);
        }
        a.t[k].p = t.param; a.t[k].g = t.grad; a.t[k].m = t.exp_avg; a.t[k].v = t.exp_avg_sq; a.t[k].n = t.n;
        a.t[k].step_size = dev ? 0.f : (float)(t.lr / (1.0 - pow(beta1, (double)t.step)));
        a.t[k].bias2_sqrt = dev ? 1.f : (float)sqrt(1.0 - pow(beta2, (double)t.step));
        a.t[k].first_block = (uint32_t)blocks; a.t[k].pad = (uint32_t)(c - c0);
        const int64_t n4 = t.n >> 2;
        const int64_t per_block = (int64_t)ADAM_BLOCK * (row_floats ? 1 : ADAM_ITEMS);       // 16-byte groups per workgroup
        blocks += n4 > 0 ? (uint64_t)((n4 + per_block - 1) / per_block) : 1;
        k++;
    }
    if (blocks > 0x7fffffffull) { gft_fail("%s: too many elements for one launch", who); return -1; }
    a.count = k;
    *blocks_out = blocks;
    return k;
}

extern "C" int gft_adam_step_rows(void* hip_stream, int32_t count, const gft_adam_tensor* tensors, int64_t rows,
                                  const uint8_t* row_mask, double beta1, double beta2, double eps, double weight_decay)
{
    if (count < 0) return gft_fail("gft_adam_step_rows: count < 0");
    if (count == 0 || rows == 0) return 0;
    if (!tensors || !row_mask || rows < 0) return gft_fail("gft_adam_step_rows: bad argument");
    for (int32_t c0 = 0; c0 < count; c0 += GFT_ADAM_MAX_TENSORS) {
        AdamRowsArgs a;
        a.row_mask = row_mask;
        uint64_t blocks = 0;
        const int k = adam_table(a.m, tensors, c0, count, beta1, beta2, eps, weight_decay, &blocks, rows, a.row_floats, "gft_adam_step_rows");
        if (k < 0) return 1;
        if (k == 0) continue;
        hipLaunchKernelGGL(k_adam_rows, dim3((unsigned)blocks), dim3(ADAM_BLOCK), 0, (hipStream_t)hip_stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return gft_fail("gft_adam_step_rows: %s", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int gft_adam_step_multi(void* hip_stream, int32_t count, const gft_adam_tensor* tensors, double beta1,
                                   double beta2, double eps, double weight_decay)
{
    if (count < 0) return gft_fail("gft_adam_step_multi: count < 0");
    if (count == 0) return 0;
    if (!tensors) return gft_fail("gft_adam_step_multi: tensors is NULL");
    for (int32_t c0 = 0; c0 < count; c0 += GFT_ADAM_MAX_TENSORS) {
        AdamMultiArgs a;
        uint64_t blocks = 0;
        const int k = adam_table(a, tensors, c0, count, beta1, beta2, eps, weight_decay, &blocks, 0, nullptr, "gft_adam_step_multi");
        if (k < 0) return 1;
        if (k == 0) continue;
        hipLaunchKernelGGL(k_adam_multi<false>, dim3((unsigned)blocks), dim3(ADAM_BLOCK), 0, (hipStream_t)hip_stream, a, (const float*)nullptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return gft_fail("gft_adam_step_multi: %s", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int gft_adam_step_multi_dev(void* hip_stream, int32_t count, const gft_adam_tensor* tensors, const double* const* lr,
                                       float* const* step, float* factors, double beta1, double beta2, double eps,
                                       double weight_decay)
{
    if (count < 0) return gft_fail("gft_adam_step_multi_dev: count < 0");
    if (count == 0) return 0;
    if (!tensors || !lr || !step || !factors) return gft_fail("gft_adam_step_multi_dev: NULL argument");
    for (int32_t c0 = 0; c0 < count; c0 += GFT_ADAM_MAX_TENSORS) {
        const int32_t c1 = count < c0 + GFT_ADAM_MAX_TENSORS ? count : c0 + GFT_ADAM_MAX_TENSORS;
        AdamTickArgs tick;
        tick.count = c1 - c0; tick.beta1 = beta1; tick.beta2 = beta2; tick.factors = factors + 2 * (size_t)c0;
        for (int32_t c = c0; c < c1; c++) {
            if (!lr[c] || !step[c]) return gft_fail("gft_adam_step_multi_dev: tensor %d: lr / step pointer is NULL", c);
            tick.lr[c - c0] = lr[c]; tick.step[c - c0] = step[c];
        }
        AdamMultiArgs a;
        uint64_t blocks = 0;
        const int k = adam_table(a, tensors, c0, count, beta1, beta2, eps, weight_decay, &blocks, 0, nullptr, "gft_adam_step_multi_dev", true);
        if (k < 0) return 1;
        // (the counts advance for every tensor of the table, as torch's capturable Adam advances state["step"] -- also for an
        // empty tensor, which takes no update)
        hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, tick);
        if (k > 0)
            hipLaunchKernelGGL(k_adam_multi<true>, dim3((unsigned)blocks), dim3(ADAM_BLOCK), 0, (hipStream_t)hip_stream, a,
                               (const float*)(factors + 2 * (size_t)c0));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return gft_fail("gft_adam_step_multi_dev: %s", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int gft_adam_step(void* hip_stream, int64_t n, float* param, const float* grad, float* exp_avg,
                             float* exp_avg_sq, double lr, double beta1, double beta2, double eps, double weight_decay,
                             int64_t step)
{
    if (n < 0) return gft_fail("gft_adam_step: n < 0");
    if (step < 1) return gft_fail("gft_adam_step: step must be >= 1");
    if (n == 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return gft_fail("gft_adam_step: NULL pointer");
    if ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
        return gft_fail("gft_adam_step: pointers must be 16-byte aligned");
    AdamArgs a;
    a.n = n; a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq;
    // torch/optim/adam.py: bias_correction1 = 1 - beta1 ** step; step_size = lr / bias_correction1;
    // bias_correction2_sqrt = (1 - beta2 ** step) ** 0.5 -- Python floats, rounded when they meet a tensor
    a.one_m_beta1 = (float)(1.0 - beta1);
    a.beta2 = (float)beta2;
    a.one_m_beta2 = (float)(1.0 - beta2);
    a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    a.bias2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    a.eps = (float)eps;
    a.weight_decay = (float)weight_decay;
    const int64_t n4 = n >> 2;
    const int64_t blocks = n4 > 0 ? (n4 + ADAM_BLOCK - 1) / ADAM_BLOCK : 1;
    hipLaunchKernelGGL(k_adam_step, dim3((unsigned)blocks), dim3(ADAM_BLOCK), 0, (hipStream_t)hip_stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gft_fail("gft_adam_step: %s", hipGetErrorString(e));
    return 0;
}
