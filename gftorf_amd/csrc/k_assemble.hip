// k_assemble.hip -- fused assembly of the rasterizer's per-Gaussian inputs (gfx950).
//
// The reference builds the seven input tensors of every render with 7 zero-filled
// allocations and 14 boolean-mask assignments in eager PyTorch
// (gaussian_renderer/__init__.py:81-105): static Gaussians (motion_mask false) are copied,
// dynamic ones get the deformation network's offsets added (and their rotation re-normalised),
// regions that are not rendered stay zero.  Each masked assignment is a nonzero() with a host
// sync, a gather and a scatter.  Here it is one pass over the data forward and one backward:
//
//   k_assemble_rank : rank of every dynamic Gaussian among the dynamic ones (row of the d_*
//                     tensors): popcount per 1024 rows, the last workgroup scans the block sums
//   k_assemble_rows : one lane per Gaussian: rank, xyz (+d_xyz), screen-space point, opacity,
//                     scale, rotation (copy | normalize(raw + d_rot))            [~110 B/Gaussian]
//   k_assemble_wide : flat over 16-byte pieces of the SH rows: f (+ d_sh[rank]) | 0
//                     (coalesced; 320 B read + 320 B written per Gaussian at M = 16)
//   k_assemble_rows_bwd / k_assemble_wide_bwd : the adjoints, written in full (zeros where no
//                     gradient flows), so the caller allocates with empty()
//
// HBM-bound elementwise work: no LDS tiling, no MFMA; 16-byte accesses, one pass.
#include "gft_internal.h"
#include "gftorf_assemble.h"

namespace {

#define ASM_BLOCK 256
#define ASM_RANK_ROWS 1024          // rows per workgroup of the rank pass
#define ASM_STATIC 0xffffffffu      // dyn_rank of a static row
#define ASM_OOB 0x80000000u         // flag on the rank of a dynamic row that has no row in the offset tensors

__global__ __launch_bounds__(ASM_RANK_ROWS) void k_assemble_rank(int P, const uint8_t* __restrict__ mask,
                                                                 uint32_t* block_sums, uint32_t* ticket,
                                                                 uint32_t* __restrict__ num_dynamic)
{
    __shared__ uint32_t s_w[ASM_RANK_ROWS / 64];
    __shared__ uint32_t s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * ASM_RANK_ROWS + tid;
    const bool m = i < P && mask[i] != 0;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(m);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0;
        for (int w = 0; w < ASM_RANK_ROWS / 64; w++) t += s_w[w];
        // device-scope store, then the ticket: the last workgroup reads the sums with device-scope loads
        __hip_atomic_store(&block_sums[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!s_last) return;
    // exclusive scan of the block sums by this workgroup (in place)
    __shared__ uint32_t s_carry;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const int nb = (int)gridDim.x;
    for (int base = 0; base < nb; base += ASM_RANK_ROWS) {
        const int b = base + tid;
        const uint32_t v = b < nb ? __hip_atomic_load(&block_sums[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += s_w[w];
        const uint32_t carry = s_carry;
        if (b < nb) block_sums[b] = carry + woff + x - v;
        __syncthreads();
        if (tid == ASM_RANK_ROWS - 1) s_carry = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) { *num_dynamic = s_carry; *ticket = 0; }
}

struct RowsArgs {
    int P;
    int render_static, render_dynamic;
    gft_assemble_io io;
    uint32_t* dyn_rank;             // [P] rank among the dynamic rows, 0xffffffff for static rows
    const uint32_t* block_sums;     // exclusive, per 1024 rows
};

// one lane per Gaussian, one workgroup per rank block of 1024 rows
__global__ __launch_bounds__(ASM_RANK_ROWS) void k_assemble_rows(RowsArgs a)
{
    __shared__ uint32_t s_w[ASM_RANK_ROWS / 64];
    const int i = blockIdx.x * ASM_RANK_ROWS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in = i < a.P;
    const bool m = in && a.io.motion_mask[i] != 0;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(m);
    if (lane == 0) s_w[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t rank = a.block_sums[blockIdx.x];
    for (int w = 0; w < wave; w++) rank += s_w[w];
    rank += (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    if (!in) return;
    // A dynamic row whose rank lies beyond the rows of the offset tensors (the reference's masked assignment raises
    // for such shapes, gaussian_renderer/__init__.py:90-104): nothing is read or written out of bounds, the row's
    // outputs are NaN (the loss shows it), its rank carries ASM_OOB so that the backward skips it too.
    const bool oob = m && (int64_t)rank >= a.io.num_offset_rows;
    a.dyn_rank[i] = m ? (oob ? (rank | ASM_OOB) : rank) : ASM_STATIC;
    const bool on = m ? a.render_dynamic != 0 : a.render_static != 0;
    float3 x = make_float3(0.f, 0.f, 0.f), s2 = x, sc = x;
    float op = 0.f;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
        x = make_float3(a.io.xyz[3 * i], a.io.xyz[3 * i + 1], a.io.xyz[3 * i + 2]);
        s2 = make_float3(a.io.screenspace[3 * i], a.io.screenspace[3 * i + 1], a.io.screenspace[3 * i + 2]);
        op = a.io.opacity[i];
        sc = make_float3(a.io.scaling[3 * i], a.io.scaling[3 * i + 1], a.io.scaling[3 * i + 2]);
        // the model's own tensors: pc.get_opacity = sigmoid(_opacity), pc.get_scaling = exp(_scaling) (scene/gaussian_model.py:
        // 123-131), in torch's expressions
        if (a.io.opacity_is_raw) op = 1.0f / (1.0f + expf(-op));
        if (a.io.scaling_is_raw) sc = make_float3(expf(sc.x), expf(sc.y), expf(sc.z));
        if (oob) {
            const float nan = __builtin_nanf("");
            x = make_float3(nan, nan, nan);
            q = make_float4(nan, nan, nan, nan);
        } else if (m) {
            if (a.io.d_xyz) {
                x.x += a.io.d_xyz[3 * (size_t)rank]; x.y += a.io.d_xyz[3 * (size_t)rank + 1]; x.z += a.io.d_xyz[3 * (size_t)rank + 2];
            } else {
                x.x += a.io.d_xyz_scalar; x.y += a.io.d_xyz_scalar; x.z += a.io.d_xyz_scalar;
            }
            float4 r = reinterpret_cast<const float4*>(a.io.rotation_raw)[i];
            if (a.io.d_rot) {
                const float4 d = reinterpret_cast<const float4*>(a.io.d_rot)[rank];
                r.x += d.x; r.y += d.y; r.z += d.z; r.w += d.w;
            } else {
                r.x += a.io.d_rot_scalar; r.y += a.io.d_rot_scalar; r.z += a.io.d_rot_scalar; r.w += a.io.d_rot_scalar;
            }
            // torch.nn.functional.normalize: v / max(||v||_2, 1e-12)
            const float nrm = fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
            q = make_float4(r.x / nrm, r.y / nrm, r.z / nrm, r.w / nrm);
        } else if (a.io.rotation) {
            q = reinterpret_cast<const float4*>(a.io.rotation)[i];
        } else {
            // no activated rotations given: pc.get_rotation = normalize(pc._rotation) (scene/gaussian_model.py) done here
            const float4 r = reinterpret_cast<const float4*>(a.io.rotation_raw)[i];
            const float nrm = fmaxf(sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w), 1e-12f);
            q = make_float4(r.x / nrm, r.y / nrm, r.z / nrm, r.w / nrm);
        }
    }
    a.io.out_means3D[3 * i] = x.x; a.io.out_means3D[3 * i + 1] = x.y; a.io.out_means3D[3 * i + 2] = x.z;
    a.io.out_means2D[3 * i] = s2.x; a.io.out_means2D[3 * i + 1] = s2.y; a.io.out_means2D[3 * i + 2] = s2.z;
    a.io.out_opacity[i] = op;
    a.io.out_scales[3 * i] = sc.x; a.io.out_scales[3 * i + 1] = sc.y; a.io.out_scales[3 * i + 2] = sc.z;
    reinterpret_cast<float4*>(a.io.out_rotations)[i] = q;
}

// out[row][c] = rendered(row) ? f[row][c] + (dynamic ? d[rank][c] | d_scalar : 0) : 0, c < row_len floats.
// VEC = 4: row_len % 4 == 0, 16-byte pieces; VEC = 1: any row length.
template <int VEC>
__global__ __launch_bounds__(ASM_BLOCK) void k_assemble_wide(size_t total_vec, int row_vec, const float* __restrict__ f,
                                                            const float* __restrict__ d, float d_scalar,
                                                            const uint32_t* __restrict__ dyn_rank, int render_static,
                                                            int render_dynamic, float* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (e >= total_vec) return;
    const size_t row = e / (size_t)row_vec;
    const int col = (int)(e - row * (size_t)row_vec);
    const uint32_t rank = dyn_rank[row];
    const bool m = rank != ASM_STATIC;
    const bool on = m ? render_dynamic != 0 : render_static != 0;
    if (m && (rank & ASM_OOB)) {              // no row in the offset tensors: see k_assemble_rows
        const float nan = __builtin_nanf("");
        if (VEC == 4) reinterpret_cast<float4*>(out)[e] = make_float4(nan, nan, nan, nan);
        else out[e] = nan;
        return;
    }
    if (VEC == 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
            v = reinterpret_cast<const float4*>(f)[e];
            if (m) {
                if (d) {
                    const float4 t = reinterpret_cast<const float4*>(d)[(size_t)rank * row_vec + col];
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                } else {
                    v.x += d_scalar; v.y += d_scalar; v.z += d_scalar; v.w += d_scalar;
                }
            }
        }
        reinterpret_cast<float4*>(out)[e] = v;
    } else {
        float v = 0.f;
        if (on) {
            v = f[e];
            if (m) v += d ? d[(size_t)rank * row_vec + col] : d_scalar;
        }
        out[e] = v;
    }
}

// The SH rows read from / handed back to the tensors the model keeps (scene/gaussian_model.py:141-153):
//   kind 0, colour: cat((dc [P,1,3], rest [P,M-1,3]), dim=1)                           a0 = dc, a1 = rest
//   kind 1, phasor: cat((cat((phase_dc, phase_rest), 1), cat((amp_dc, amp_rest), 1)), dim=2)  a = phase, b = amp; [P,1,1] / [P,M-1,1]
template <typename T>
struct SplitPtrs { T* a0; T* a1; T* b0; T* b1; int M; int kind; };
template <typename T>
__device__ __forceinline__ T* split_at(const SplitPtrs<T>& s, size_t row, int c)
{
    if (s.kind == 0) return c < 3 ? s.a0 + row * 3 + c : s.a1 + row * (size_t)(3 * (s.M - 1)) + (c - 3);
    const int k = c >> 1;
    T* dc = (c & 1) ? s.b0 : s.a0;
    T* rest = (c & 1) ? s.b1 : s.a1;
    return k == 0 ? dc + row : rest + row * (size_t)(s.M - 1) + (k - 1);
}

// k_assemble_wide with the feature row gathered from its parts (VEC floats per thread: 4 when the row allows, else 1)
template <int VEC>
__global__ __launch_bounds__(ASM_BLOCK) void k_assemble_wide_split(size_t total_vec, int row_vec, SplitPtrs<const float> f,
                                                                  const float* __restrict__ d, float d_scalar,
                                                                  const uint32_t* __restrict__ dyn_rank, int render_static,
                                                                  int render_dynamic, float* __restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (e >= total_vec) return;
    const size_t row = e / (size_t)row_vec;
    const int col = (int)(e - row * (size_t)row_vec);
    const uint32_t rank = dyn_rank[row];
    const bool m = rank != ASM_STATIC;
    const bool on = m ? render_dynamic != 0 : render_static != 0;
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k++) v[k] = 0.f;
    if (m && (rank & ASM_OOB)) {              // no row in the offset tensors: see k_assemble_rows
#pragma unroll
        for (int k = 0; k < VEC; k++) v[k] = __builtin_nanf("");
    } else if (on) {
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            v[k] = *split_at(f, row, VEC * col + k);
            if (m) v[k] += d ? d[((size_t)rank * row_vec + col) * VEC + k] : d_scalar;
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; k++) out[e * VEC + k] = v[k];
}

// k_assemble_wide_bwd with the features' gradient scattered to its parts
template <int VEC>
__global__ __launch_bounds__(ASM_BLOCK) void k_assemble_wide_bwd_split(size_t total_vec, int row_vec, const float* __restrict__ g,
                                                                      const uint32_t* __restrict__ dyn_rank, int render_static,
                                                                      int render_dynamic, SplitPtrs<float> g_f, float* __restrict__ g_d)
{
    const size_t e = (size_t)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (e >= total_vec) return;
    const size_t row = e / (size_t)row_vec;
    const int col = (int)(e - row * (size_t)row_vec);
    const uint32_t rank_w = dyn_rank[row];
    const bool oob = rank_w != ASM_STATIC && (rank_w & ASM_OOB);
    const bool m = rank_w != ASM_STATIC && !oob;
    const bool on = (rank_w != ASM_STATIC ? render_dynamic != 0 : render_static != 0) && !oob;
#pragma unroll
    for (int k = 0; k < VEC; k++) {
        const float v = (on && g) ? g[e * VEC + k] : 0.f;
        float* dst = split_at(g_f, row, VEC * col + k);
        // (a part nobody wants: its base pointer is NULL -- test the base, not the element's address)
        const float* base = g_f.kind == 0 ? (VEC * col + k < 3 ? g_f.a0 : g_f.a1)
                                          : (((VEC * col + k) & 1) ? ((VEC * col + k) >> 1 ? g_f.b1 : g_f.b0) : ((VEC * col + k) >> 1 ? g_f.a1 : g_f.a0));
        if (base) *dst = v;
        if (m && g_d) g_d[((size_t)rank_w * row_vec + col) * VEC + k] = v;
    }
}

struct RowsBwdArgs {
    int P;
    int render_static, render_dynamic;
    gft_assemble_bwd_io io;
    const uint32_t* dyn_rank;
};

__global__ __launch_bounds__(ASM_BLOCK) void k_assemble_rows_bwd(RowsBwdArgs a)
{
    const int i = blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (i >= a.P) return;
    const uint32_t rank_w = a.dyn_rank[i];
    const bool m = rank_w != ASM_STATIC;
    const bool oob = m && (rank_w & ASM_OOB);   // no row in the offset tensors (k_assemble_rows): zero gradients
    const uint32_t rank = rank_w & ~ASM_OOB;
    const bool on = (m ? a.render_dynamic != 0 : a.render_static != 0) && !oob;
    float3 gx = make_float3(0.f, 0.f, 0.f), gs = gx, gsc = gx;
    float gop = 0.f;
    float4 gq = make_float4(0.f, 0.f, 0.f, 0.f), gq_raw = gq;
    if (on) {
        if (a.io.g_means3D) gx = make_float3(a.io.g_means3D[3 * i], a.io.g_means3D[3 * i + 1], a.io.g_means3D[3 * i + 2]);
        if (a.io.g_means2D) gs = make_float3(a.io.g_means2D[3 * i], a.io.g_means2D[3 * i + 1], a.io.g_means2D[3 * i + 2]);
        if (a.io.g_opacity) gop = a.io.g_opacity[i];
        if (a.io.g_scales) gsc = make_float3(a.io.g_scales[3 * i], a.io.g_scales[3 * i + 1], a.io.g_scales[3 * i + 2]);
        const float4 g = a.io.g_rotations ? reinterpret_cast<const float4*>(a.io.g_rotations)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (m || a.io.static_from_raw) {
            float4 r = reinterpret_cast<const float4*>(a.io.rotation_raw)[i];
            if (!m) {
            } else if (a.io.d_rot) {
                const float4 d = reinterpret_cast<const float4*>(a.io.d_rot)[rank];
                r.x += d.x; r.y += d.y; r.z += d.z; r.w += d.w;
            } else {
                r.x += a.io.d_rot_scalar; r.y += a.io.d_rot_scalar; r.z += a.io.d_rot_scalar; r.w += a.io.d_rot_scalar;
            }
            const float n2 = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
            if (n2 > 1e-12f) {
                // y = v / n: dv = (g - y (y . g)) / n
                const float inv = 1.0f / n2;
                const float4 y = make_float4(r.x * inv, r.y * inv, r.z * inv, r.w * inv);
                const float dot = y.x * g.x + y.y * g.y + y.z * g.z + y.w * g.w;
                gq_raw = make_float4((g.x - y.x * dot) * inv, (g.y - y.y * dot) * inv, (g.z - y.z * dot) * inv,
                                     (g.w - y.w * dot) * inv);
            } else {
                // clamped denominator: y = v / 1e-12
                gq_raw = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
            }
        } else {
            gq = g;
        }
    }
    if (on && a.io.opacity_raw) {          // sigmoid_backward: g (1 - y) y
        const float y = 1.0f / (1.0f + expf(-a.io.opacity_raw[i]));
        gop = gop * ((1.0f - y) * y);
    }
    if (on && a.io.scaling_raw) {          // exp's backward: g y
        gsc.x *= expf(a.io.scaling_raw[3 * i]); gsc.y *= expf(a.io.scaling_raw[3 * i + 1]); gsc.z *= expf(a.io.scaling_raw[3 * i + 2]);
    }
    if (a.io.g_xyz) { a.io.g_xyz[3 * i] = gx.x; a.io.g_xyz[3 * i + 1] = gx.y; a.io.g_xyz[3 * i + 2] = gx.z; }
    if (a.io.g_screenspace) { a.io.g_screenspace[3 * i] = gs.x; a.io.g_screenspace[3 * i + 1] = gs.y; a.io.g_screenspace[3 * i + 2] = gs.z; }
    if (a.io.g_opacity_in) a.io.g_opacity_in[i] = gop;
    if (a.io.g_scaling) { a.io.g_scaling[3 * i] = gsc.x; a.io.g_scaling[3 * i + 1] = gsc.y; a.io.g_scaling[3 * i + 2] = gsc.z; }
    if (a.io.g_rotation) reinterpret_cast<float4*>(a.io.g_rotation)[i] = gq;
    if (a.io.g_rotation_raw) reinterpret_cast<float4*>(a.io.g_rotation_raw)[i] = gq_raw;
    if (m && !oob) {
        // rows of the offset tensors: every dynamic row is written (zeros when the region is off)
        if (a.io.g_d_xyz) {
            a.io.g_d_xyz[3 * (size_t)rank] = gx.x; a.io.g_d_xyz[3 * (size_t)rank + 1] = gx.y; a.io.g_d_xyz[3 * (size_t)rank + 2] = gx.z;
        }
        if (a.io.g_d_rot) reinterpret_cast<float4*>(a.io.g_d_rot)[rank] = gq_raw;
    }
}

// g_f[row][c] = rendered(row) ? g[row][c] : 0; g_d[rank][c] = g[row][c] for dynamic rendered rows (0 if off)
template <int VEC>
__global__ __launch_bounds__(ASM_BLOCK) void k_assemble_wide_bwd(size_t total_vec, int row_vec, const float* __restrict__ g,
                                                                const uint32_t* __restrict__ dyn_rank, int render_static,
                                                                int render_dynamic, float* __restrict__ g_f,
                                                                float* __restrict__ g_d)
{
    const size_t e = (size_t)blockIdx.x * ASM_BLOCK + threadIdx.x;
    if (e >= total_vec) return;
    const size_t row = e / (size_t)row_vec;
    const int col = (int)(e - row * (size_t)row_vec);
    const uint32_t rank_w = dyn_rank[row];
    const bool oob = rank_w != ASM_STATIC && (rank_w & ASM_OOB);
    const bool m = rank_w != ASM_STATIC && !oob;
    const uint32_t rank = rank_w;
    const bool on = (rank_w != ASM_STATIC ? render_dynamic != 0 : render_static != 0) && !oob;
    // (g_f == NULL: the caller hands `g` itself on as the features' gradient -- with both regions rendered the two are the
    // same values -- and only the dynamic rows are read, for g_d)
    if (!g_f && !(m && g_d)) return;
    if (VEC == 4) {
        const float4 v = (on && g) ? reinterpret_cast<const float4*>(g)[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_f) reinterpret_cast<float4*>(g_f)[e] = v;
        if (m && g_d) reinterpret_cast<float4*>(g_d)[(size_t)rank * row_vec + col] = v;
    } else {
        const float v = (on && g) ? g[e] : 0.f;
        if (g_f) g_f[e] = v;
        if (m && g_d) g_d[(size_t)rank * row_vec + col] = v;
    }
}

// g_d alone (the features' gradient is `g` itself: see k_assemble_wide_bwd): four threads per row, a dynamic row's 16-byte
// pieces dealt round-robin to them (64 contiguous bytes per four threads and trip); the threads of a static row leave
// after one 4-byte read.  30 % dynamic rows at 1 M Gaussians: 47 -> 2x us against the flat kernel with its copy switched off.
__global__ __launch_bounds__(ASM_BLOCK) void k_assemble_wide_gather(int P, int row_vec, const float4* __restrict__ g,
                                                                   const uint32_t* __restrict__ dyn_rank, int render_dynamic,
                                                                   float4* __restrict__ g_d)
{
    const size_t t = (size_t)blockIdx.x * ASM_BLOCK + threadIdx.x;
    const size_t row = t >> 2;
    if (row >= (size_t)P) return;
    const uint32_t rank_w = dyn_rank[row];
    if (rank_w == ASM_STATIC || (rank_w & ASM_OOB)) return;
    const bool on = render_dynamic != 0 && g != nullptr;
    for (int col = (int)(t & 3); col < row_vec; col += 4)
        g_d[(size_t)rank_w * row_vec + col] = on ? g[row * (size_t)row_vec + col] : make_float4(0.f, 0.f, 0.f, 0.f);
}

int launch_wide(hipStream_t s, int P, int row_floats, const float* f, const float* d, float d_scalar,
                const uint32_t* rank, int rs, int rd, float* out)
{
    if (!f || !out || row_floats <= 0 || P <= 0) return 0;
    if (row_floats % 4 == 0) {
        const size_t total = (size_t)P * (row_floats / 4);
        hipLaunchKernelGGL(k_assemble_wide<4>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s, total,
                           row_floats / 4, f, d, d_scalar, rank, rs, rd, out);
    } else {
        const size_t total = (size_t)P * row_floats;
        hipLaunchKernelGGL(k_assemble_wide<1>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s, total,
                           row_floats, f, d, d_scalar, rank, rs, rd, out);
    }
    return 0;
}

int launch_wide_bwd(hipStream_t s, int P, int row_floats, const float* g, const uint32_t* rank, int rs, int rd,
                    float* g_f, float* g_d)
{
    if ((!g_f && !g_d) || row_floats <= 0 || P <= 0) return 0;
    if (!g_f && row_floats % 4 == 0) {
        hipLaunchKernelGGL(k_assemble_wide_gather, dim3((unsigned)(((size_t)P * 4 + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s, P,
                           row_floats / 4, reinterpret_cast<const float4*>(g), rank, rd, reinterpret_cast<float4*>(g_d));
        return 0;
    }
    if (row_floats % 4 == 0) {
        const size_t total = (size_t)P * (row_floats / 4);
        hipLaunchKernelGGL(k_assemble_wide_bwd<4>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s,
                           total, row_floats / 4, g, rank, rs, rd, g_f, g_d);
    } else {
        const size_t total = (size_t)P * row_floats;
        hipLaunchKernelGGL(k_assemble_wide_bwd<1>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s,
                           total, row_floats, g, rank, rs, rd, g_f, g_d);
    }
    return 0;
}

int launch_wide_split(hipStream_t s, int P, int row_floats, const SplitPtrs<const float>& f, const float* d, float d_scalar,
                      const uint32_t* rank, int rs, int rd, float* out)
{
    if (!f.a0 || !out || row_floats <= 0 || P <= 0) return 0;
    if (row_floats % 4 == 0) {
        const size_t total = (size_t)P * (row_floats / 4);
        hipLaunchKernelGGL(k_assemble_wide_split<4>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s, total,
                           row_floats / 4, f, d, d_scalar, rank, rs, rd, out);
    } else {
        const size_t total = (size_t)P * row_floats;
        hipLaunchKernelGGL(k_assemble_wide_split<1>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s, total,
                           row_floats, f, d, d_scalar, rank, rs, rd, out);
    }
    return 0;
}

int launch_wide_bwd_split(hipStream_t s, int P, int row_floats, const float* g, const uint32_t* rank, int rs, int rd,
                          const SplitPtrs<float>& g_f, float* g_d)
{
    if (row_floats <= 0 || P <= 0) return 0;
    if (row_floats % 4 == 0) {
        const size_t total = (size_t)P * (row_floats / 4);
        hipLaunchKernelGGL(k_assemble_wide_bwd_split<4>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s,
                           total, row_floats / 4, g, rank, rs, rd, g_f, g_d);
    } else {
        const size_t total = (size_t)P * row_floats;
        hipLaunchKernelGGL(k_assemble_wide_bwd_split<1>, dim3((unsigned)((total + ASM_BLOCK - 1) / ASM_BLOCK)), dim3(ASM_BLOCK), 0, s,
                           total, row_floats, g, rank, rs, rd, g_f, g_d);
    }
    return 0;
}

}  // namespace

extern "C" size_t gft_assemble_scratch_bytes(int32_t P)
{
    const size_t p = (size_t)(P > 0 ? P : 0);
    // dyn_rank u32[P] | block sums u32[ceil(P/1024)] | ticket, num_dynamic
    return p * 4 + ((p + ASM_RANK_ROWS - 1) / ASM_RANK_ROWS) * 4 + 256;
}

static void assemble_scratch(void* scratch, int32_t P, uint32_t** rank, uint32_t** sums, uint32_t** ticket, uint32_t** ndyn)
{
    char* b = (char*)scratch;
    const size_t p = (size_t)P;
    *rank = (uint32_t*)b;
    *sums = (uint32_t*)(b + p * 4);
    uint32_t* tail = (uint32_t*)(b + p * 4 + ((p + ASM_RANK_ROWS - 1) / ASM_RANK_ROWS) * 4);
    // 16-byte aligned tail
    tail = (uint32_t*)(((uintptr_t)tail + 15) & ~(uintptr_t)15);
    *ticket = tail;
    *ndyn = tail + 1;
}

extern "C" int gft_assemble_forward(void* hip_stream, int32_t P, int32_t M, int32_t M_p, int32_t render_static,
                                    int32_t render_dynamic, const gft_assemble_io* io)
{
    if (P < 0 || M < 0 || M_p < 0) return gft_fail("gft_assemble_forward: negative size");
    if (!io) return gft_fail("gft_assemble_forward: io is NULL");
    if (P == 0) return 0;
    if (!io->xyz || !io->screenspace || !io->opacity || !io->scaling || !io->rotation_raw ||
        !io->motion_mask || !io->scratch || !io->out_means3D || !io->out_means2D || !io->out_opacity || !io->out_scales ||
        !io->out_rotations)
        return gft_fail("gft_assemble_forward: required pointer is NULL");
    // (out_shs / out_shs_p may be NULL with M, M_p > 0: the caller uses the feature tensor itself -- a zero offset, both regions)
    // (the feature tensors whole, or in the parts the model keeps: gft_assemble_io.feat_dc_color ...)
    const bool color_parts = !io->feat_color && io->feat_dc_color, phasor_parts = !io->feat_phasor && io->phase_dc;
    if ((M > 0) != (io->feat_color != nullptr || color_parts) || (M == 0 && io->out_shs != nullptr))
        return gft_fail("gft_assemble_forward: M does not match feat_color / out_shs");
    if ((M_p > 0) != (io->feat_phasor != nullptr || phasor_parts) || (M_p == 0 && io->out_shs_p != nullptr))
        return gft_fail("gft_assemble_forward: M_p does not match feat_phasor / out_shs_p");
    if (color_parts && M > 1 && !io->feat_rest_color) return gft_fail("gft_assemble_forward: feat_rest_color is NULL with M > 1");
    if (phasor_parts && (!io->amp_dc || (M_p > 1 && (!io->phase_rest || !io->amp_rest))))
        return gft_fail("gft_assemble_forward: the phasor features' parts are incomplete");
    if ((io->d_xyz || io->d_rot || io->d_sh || io->d_sh_p) && (io->num_offset_rows < 0 || io->num_offset_rows > P))
        return gft_fail("gft_assemble_forward: the offset tensors have %lld rows for %d Gaussians", (long long)io->num_offset_rows, P);
    hipStream_t s = (hipStream_t)hip_stream;
    uint32_t *rank, *sums, *ticket, *ndyn;
    assemble_scratch(io->scratch, P, &rank, &sums, &ticket, &ndyn);
    GFT_CHECK_HIP(gft_zero_async(ticket, 8, s));
    const int nrb = (P + ASM_RANK_ROWS - 1) / ASM_RANK_ROWS;
    hipLaunchKernelGGL(k_assemble_rank, dim3(nrb), dim3(ASM_RANK_ROWS), 0, s, P, io->motion_mask, sums, ticket, ndyn);
    RowsArgs a;
    a.P = P; a.render_static = render_static; a.render_dynamic = render_dynamic;
    a.io = *io;
    // rows of the offset tensors; with scalar offsets only, every rank is in range
    if (!io->d_xyz && !io->d_rot && !io->d_sh && !io->d_sh_p) a.io.num_offset_rows = (int64_t)1 << 40;
    a.dyn_rank = rank;
    a.block_sums = sums;
    hipLaunchKernelGGL(k_assemble_rows, dim3(nrb), dim3(ASM_RANK_ROWS), 0, s, a);
    if (color_parts)
        launch_wide_split(s, P, M * 3, SplitPtrs<const float>{io->feat_dc_color, io->feat_rest_color, nullptr, nullptr, M, 0}, io->d_sh,
                          io->d_sh_scalar, rank, render_static, render_dynamic, io->out_shs);
    else launch_wide(s, P, M * 3, io->feat_color, io->d_sh, io->d_sh_scalar, rank, render_static, render_dynamic, io->out_shs);
    if (phasor_parts)
        launch_wide_split(s, P, M_p * 2, SplitPtrs<const float>{io->phase_dc, io->phase_rest, io->amp_dc, io->amp_rest, M_p, 1}, io->d_sh_p,
                          io->d_sh_p_scalar, rank, render_static, render_dynamic, io->out_shs_p);
    else launch_wide(s, P, M_p * 2, io->feat_phasor, io->d_sh_p, io->d_sh_p_scalar, rank, render_static, render_dynamic, io->out_shs_p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gft_fail("gft_assemble_forward: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int gft_assemble_num_dynamic(void* hip_stream, int32_t P, const void* scratch, int64_t* num_dynamic)
{
    if (!num_dynamic) return gft_fail("gft_assemble_num_dynamic: NULL argument");
    *num_dynamic = 0;
    if (P <= 0) return 0;
    if (!scratch) return gft_fail("gft_assemble_num_dynamic: scratch is NULL");
    uint32_t *rank, *sums, *ticket, *ndyn;
    assemble_scratch(const_cast<void*>(scratch), P, &rank, &sums, &ticket, &ndyn);
    uint32_t host = 0;
    GFT_CHECK_HIP(hipMemcpyAsync(&host, ndyn, 4, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    GFT_CHECK_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    *num_dynamic = (int64_t)host;
    return 0;
}

extern "C" int gft_assemble_backward(void* hip_stream, int32_t P, int32_t M, int32_t M_p, int32_t render_static,
                                     int32_t render_dynamic, const gft_assemble_bwd_io* io)
{
    if (P < 0 || M < 0 || M_p < 0) return gft_fail("gft_assemble_backward: negative size");
    if (!io) return gft_fail("gft_assemble_backward: io is NULL");
    if (P == 0) return 0;
    if (!io->scratch || !io->rotation_raw) return gft_fail("gft_assemble_backward: required pointer is NULL");
    hipStream_t s = (hipStream_t)hip_stream;
    uint32_t *rank, *sums, *ticket, *ndyn;
    assemble_scratch(const_cast<void*>(io->scratch), P, &rank, &sums, &ticket, &ndyn);
    RowsBwdArgs a;
    a.P = P; a.render_static = render_static; a.render_dynamic = render_dynamic;
    a.io = *io;
    a.dyn_rank = rank;
    hipLaunchKernelGGL(k_assemble_rows_bwd, dim3((P + ASM_BLOCK - 1) / ASM_BLOCK), dim3(ASM_BLOCK), 0, s, a);
    if (io->g_feat_dc_color || io->g_feat_rest_color)
        launch_wide_bwd_split(s, P, M * 3, io->g_shs, rank, render_static, render_dynamic,
                              SplitPtrs<float>{io->g_feat_dc_color, io->g_feat_rest_color, nullptr, nullptr, M, 0}, io->g_d_sh);
    else launch_wide_bwd(s, P, M * 3, io->g_shs, rank, render_static, render_dynamic, io->g_feat_color, io->g_d_sh);
    if (io->g_phase_dc || io->g_phase_rest || io->g_amp_dc || io->g_amp_rest)
        launch_wide_bwd_split(s, P, M_p * 2, io->g_shs_p, rank, render_static, render_dynamic,
                              SplitPtrs<float>{io->g_phase_dc, io->g_phase_rest, io->g_amp_dc, io->g_amp_rest, M_p, 1}, io->g_d_sh_p);
    else launch_wide_bwd(s, P, M_p * 2, io->g_shs_p, rank, render_static, render_dynamic, io->g_feat_phasor, io->g_d_sh_p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gft_fail("gft_assemble_backward: %s", hipGetErrorString(e));
    return 0;
}
