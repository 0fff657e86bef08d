// k_deform.hip -- the deformation network (reference utils/time_utils.py:56-127) on the gfx950 fp32
// matrix cores.  See include/gftorf_deform.h for the contract.
//
// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32]; lane l holds A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31]; D: column = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).
// 64 cycles per instruction per SIMD = the fp32 peak (157 TFLOP/s), so the kernels only have to
// keep one wave per SIMD fed.  The walks keep a tile of points' activations in LDS (16-byte reads,
// 4 k per read) and stream the weights L2 -> registers, each wave its own 64 output columns (the packed
// weights are interleaved [k/4][n][4] for 16-byte loads), one 16-k chunk ahead of the multiply; the
// weight-gradient kernel reads both operands straight from global memory.
#include "gft_internal.h"
#include "gftorf_deform.h"
#include "gftorf_densify.h"

#include <cstdio>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int DF_D = GFT_DEFORM_LAYERS;
constexpr int DF_W = GFT_DEFORM_WIDTH;
// The encoded input has `in` = 3 + 6 xyz_multires + 1 + 2 t_multires columns (run-time: 84 for the reference's
// configured network, arguments/__init__.py:68-69; 76 for the class default, time_utils.py:57); every buffer holds
// DF_INK = 96 columns, the ones from `in` on are zero (zero weight rows, zero encoding columns).
constexpr int DF_INK = GFT_DEFORM_MAX_INPUTS;   // encoding as a GEMM k-extent (multiple of 16)
constexpr int DF_EMB = GFT_DEFORM_MAX_INPUTS;   // stored encoding row (3 column tiles of the weight-gradient GEMM)
static_assert(DF_INK == 96, "the encoding tiles assume 96 columns");
constexpr int DF_HEAD = 64;                // head columns: 48 (d_sh, [coefficient][channel]) + 3 (d_xyz) + pad
// 32-point row tiles per wave = points per workgroup / 32.  Measured on MI355X (300 k points): forward 64 points
// 2.70 ms (one workgroup per CU; 2.83 with two), 96 points 3.21 ms; backward 64 points 2.89 ms, 96 points 2.81 ms.
constexpr int DF_NR_FWD = 2;
constexpr int DF_NR_BWD = 3;
constexpr int DF_PAD = 192;                // point counts are padded to a multiple of both walk tiles and of 64
constexpr int DF_ES = 100;                 // LDS row stride of the encoding (16-byte reads of 16 rows hit 16 bank groups)
constexpr int DF_DW_TILE = 64;             // point granularity of the weight-gradient splits
constexpr int DF_HS = 260;                 // LDS row strides (floats): 16-byte reads of 16 rows hit 16 bank groups

// packed parameter buffer (floats): forward stream, backward stream, biases
constexpr int64_t DF_F_SZ0 = (int64_t)DF_INK * DF_W;                  // 20480
constexpr int64_t DF_F_SZ = (int64_t)DF_W * DF_W;                     // 65536
constexpr int64_t DF_F_HEAD_SZ = (int64_t)DF_W * DF_HEAD;             // 16384
// forward stream, in the order it is used: L0 | L1 L2 L3 L4 | L5 (encoding rows) | L5 (hidden rows) | L6 | L7 | heads
constexpr int64_t DF_F_TOTAL = 2 * DF_F_SZ0 + 7 * DF_F_SZ + DF_F_HEAD_SZ;   // 516096
constexpr int64_t DF_B_BASE = DF_F_TOTAL;
constexpr int64_t DF_B_TOTAL = DF_F_HEAD_SZ + 7 * DF_F_SZ;            // 475136
constexpr int64_t DF_BIAS_BASE = DF_B_BASE + DF_B_TOTAL;              // 991232
constexpr int64_t DF_PACKED_FLOATS = DF_BIAS_BASE + DF_D * DF_W + DF_HEAD;   // 993344

// bf16 plane copy of the two weight streams (after the fp32 floats): every fp32 weight as hi + mid + lo bf16
// (24 mantissa bits), per segment [plane][k/8][n][8]: the operand layout of v_mfma_f32_32x32x16_bf16
constexpr int64_t DF_BF_ELEMS = DF_F_TOTAL + DF_B_TOTAL;              // weights in both streams
constexpr int64_t DF_BF_FLOATS = DF_BF_ELEMS * 3 / 2;                 // 3 x 2 bytes each
constexpr int DF_NSEG = 18;
// stream segments: first element (in either representation's element count), rows K, columns
struct DfSeg { int64_t off; int K, ncol; };
__host__ __device__ inline DfSeg df_seg(int s)
{
    // forward: L0 | L1..L4 | L5 encoding rows | L5 hidden rows, L6, L7 | heads; backward: heads, L7 .. L1
    if (s == 0) return {0, DF_INK, DF_W};
    if (s <= 4) return {DF_F_SZ0 + (int64_t)(s - 1) * DF_F_SZ, DF_W, DF_W};
    if (s == 5) return {DF_F_SZ0 + 4 * DF_F_SZ, DF_INK, DF_W};
    if (s <= 8) return {2 * DF_F_SZ0 + (int64_t)(s - 2) * DF_F_SZ, DF_W, DF_W};
    if (s == 9) return {DF_F_TOTAL - DF_F_HEAD_SZ, DF_W, DF_HEAD};
    if (s == 10) return {DF_B_BASE, DF_HEAD, DF_W};
    return {DF_B_BASE + DF_F_HEAD_SZ + (int64_t)(s - 11) * DF_F_SZ, DF_W, DF_W};
}

// ---------------------------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    gft_deform_params p;
    float* out;
    int in;                                // encoded inputs (<= DF_INK)
    uint32_t* flag;                        // range flag of the fp16 stream, cleared here (k_deform_pack_h sets it)
};

// head column hc -> (weight row pointer, bias): columns 0..47 are d_sh[coefficient c][channel ch] = c*3+ch
__device__ __forceinline__ const float* head_row(const gft_deform_params& p, int hc, float& bias)
{
    if (hc < 48) {
        const int c = hc / 3, ch = hc - 3 * c;
        const float* w = ch == 0 ? p.r_w : ch == 1 ? p.g_w : p.b_w;
        const float* b = ch == 0 ? p.r_b : ch == 1 ? p.g_b : p.b_b;
        bias = b[c];
        return w + (size_t)c * DF_W;
    }
    if (hc < 51) {
        bias = p.xyz_b[hc - 48];
        return p.xyz_w + (size_t)(hc - 48) * DF_W;
    }
    bias = 0.f;
    return nullptr;
}

__global__ __launch_bounds__(256) void k_deform_pack(PackArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= DF_PACKED_FLOATS) return;
    if (e == 0) a.flag[0] = 0u;
    float v = 0.f;
    if (e < DF_F_TOTAL) {
        // forward stream: per segment [k/4][ncol][4], element (k, n) = W[n][k]
        const int64_t enc5 = DF_F_SZ0 + 4 * DF_F_SZ;          // start of layer 5's encoding rows
        if (e < DF_F_SZ0 || (e >= enc5 && e < enc5 + DF_F_SZ0)) {
            const int l = e < DF_F_SZ0 ? 0 : 5;
            const int64_t r = e < DF_F_SZ0 ? e : e - enc5;
            const int kq = (int)(r / (DF_W * 4)), n = (int)((r >> 2) % DF_W), k = 4 * kq + (int)(r & 3);
            const int ld = l == 0 ? a.in : DF_W + a.in;
            v = k < a.in ? a.p.linear_w[l][(size_t)n * ld + k] : 0.f;
        } else if (e < DF_F_TOTAL - DF_F_HEAD_SZ) {
            const int64_t r1 = e < enc5 ? e - DF_F_SZ0 : e - 2 * DF_F_SZ0;
            const int l = 1 + (int)(r1 / DF_F_SZ);
            const int64_t r = r1 % DF_F_SZ;
            const int kq = (int)(r / (DF_W * 4)), n = (int)((r >> 2) % DF_W), k = 4 * kq + (int)(r & 3);
            v = l == 5 ? a.p.linear_w[5][(size_t)n * (DF_W + a.in) + a.in + k] : a.p.linear_w[l][(size_t)n * DF_W + k];
        } else {
            const int64_t r = e - (DF_F_TOTAL - DF_F_HEAD_SZ);
            const int kq = (int)(r / (DF_HEAD * 4)), n = (int)((r >> 2) % DF_HEAD), k = 4 * kq + (int)(r & 3);
            float bias;
            const float* row = head_row(a.p, n, bias);
            v = row ? row[k] : 0.f;
        }
    } else if (e < DF_BIAS_BASE) {
        // backward stream: head first, then layers 7..1; element (k = output, n = input) = W[k][n]
        const int64_t r0 = e - DF_B_BASE;
        if (r0 < DF_F_HEAD_SZ) {
            const int kq = (int)(r0 / (DF_W * 4)), n = (int)((r0 >> 2) % DF_W), k = 4 * kq + (int)(r0 & 3);
            float bias;
            const float* row = head_row(a.p, k, bias);
            v = row ? row[n] : 0.f;
        } else {
            const int64_t r1 = r0 - DF_F_HEAD_SZ;
            const int l = 7 - (int)(r1 / DF_F_SZ);
            const int64_t r = r1 % DF_F_SZ;
            const int kq = (int)(r / (DF_W * 4)), n = (int)((r >> 2) % DF_W), k = 4 * kq + (int)(r & 3);
            v = l == 5 ? a.p.linear_w[5][(size_t)k * (DF_W + a.in) + a.in + n] : a.p.linear_w[l][(size_t)k * DF_W + n];
        }
    } else {
        const int r = (int)(e - DF_BIAS_BASE);
        if (r < DF_D * DF_W) v = a.p.linear_b[r >> 8][r & 255];
        else (void)head_row(a.p, r - DF_D * DF_W, v);
    }
    a.out[e] = v;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// x = hi + mid + lo to 24 bits: every product of two such numbers is the sum of nine exact bf16 products
__device__ __forceinline__ void split3(float x, __bf16& hi, __bf16& mid, __bf16& lo)
{
    hi = (__bf16)x;
    const float r1 = x - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two floats -> two bf16 (round to nearest even, like the scalar casts of split3) in one dword, the first in the low half
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b)
{
    const f32x2_t v = {a, b};
    const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}

// fp32 packed streams ([k/4][n][4] per segment) -> bf16 planes ([plane][k/8][n][8] per segment)
__global__ __launch_bounds__(256) void k_deform_pack_bf(const float* __restrict__ packed, __bf16* __restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= DF_BF_ELEMS) return;
    int sgi = 0;
    for (int q = 1; q < DF_NSEG; q++)
        if (e >= df_seg(q).off) sgi = q;
    const DfSeg sg = df_seg(sgi);
    const int64_t r = e - sg.off;
    const int kq = (int)(r / (sg.ncol * 4)), n = (int)((r >> 2) % sg.ncol), k = 4 * kq + (int)(r & 3);
    __bf16 hi, mid, lo;
    split3(packed[e], hi, mid, lo);
    const int64_t plane = (int64_t)sg.K * sg.ncol;
    const int64_t o = 3 * sg.off + ((int64_t)(k >> 3) * sg.ncol + n) * 8 + (k & 7);
    out[o] = hi;
    out[o + plane] = mid;
    out[o + 2 * plane] = lo;
}

// ---------------------------------------------------------------------------------------------
// the streamed GEMM both walks use: acc[rt][ct] += (A[64 x K] * W[K x ncol])^T, 16 k per chunk.
// The weights are the MFMA's A operand and the activations its B operand, so a result tile has the POINT on
// the lane and the output column on the registers: registers 4g..4g+3 of lane (j, hh) are columns
// 8g + 4hh .. +3 of point j -- 16-byte LDS and global stores in the epilogues.
// ---------------------------------------------------------------------------------------------
template <int NR, int NC>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NR][NC])
{
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[r][c][q] = 0.f;
}

__device__ __forceinline__ float f4_get(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

// A wave owns its columns of W: no other wave reads them, so the weights go L2 -> registers directly
// (lane (j, hh) reads the float4 [k/4 = 2r + hh][column j]: 512 contiguous bytes per half wave), one
// chunk ahead of the multiply; the A tile (activations, shared by the four waves) is read from LDS, one
// chunk ahead as well.  No barrier inside a layer.
// k order inside a chunk: (0,4) (1,5) (2,6) (3,7) (8,12) ... -- any fixed order is a valid fp32 sum.
template <int NC, int NCOL>
__device__ __forceinline__ void load_w(float4 (&w)[2][NC], const float4* wlane)
{
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++) w[r][ct] = wlane[2 * r * NCOL + ct * 32];
}

// wlane: the lane's float4 of the segment's first chunk ([hh][first column + j]); wcur: that chunk, already
// loaded.  more_after: the stream continues with a chunk of the same shape (it is prefetched into wcur).
template <int NR, int NC, int NCOL>
__device__ __forceinline__ void stream_gemm(f32x16 (&acc)[NR][NC], const float* a_lane, int a_stride, int nchunks,
                                            const float4*& wlane, float4 (&wcur)[2][NC], bool more_after)
{
    float4 acur[2][NR];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int rt = 0; rt < NR; rt++) acur[r][rt] = *reinterpret_cast<const float4*>(a_lane + rt * 32 * a_stride + 8 * r);
    for (int c = 0; c < nchunks; c++) {
        const bool last = c + 1 >= nchunks;
        // (unconditional loads from clamped addresses: conditionally filled arrays end up in scratch)
        const float4* wn = wlane + ((!last || more_after) ? 4 * NCOL : 0);
        const float* an = a_lane + (last ? c : c + 1) * 16;
        float4 wnxt[2][NC], anxt[2][NR];
        load_w<NC, NCOL>(wnxt, wn);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int rt = 0; rt < NR; rt++) anxt[r][rt] = *reinterpret_cast<const float4*>(an + rt * 32 * a_stride + 8 * r);
        // keep the loads above the multiply (the scheduler otherwise sinks them to their use, after it)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int rt = 0; rt < NR; rt++)
#pragma unroll
                    for (int ct = 0; ct < NC; ct++)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4_get(wcur[r][ct], s), f4_get(acur[r][rt], s),
                                                                           acc[rt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        wlane = wn;
#pragma unroll
        for (int r = 0; r < 2; r++) {
#pragma unroll
            for (int ct = 0; ct < NC; ct++) wcur[r][ct] = wnxt[r][ct];
#pragma unroll
            for (int rt = 0; rt < NR; rt++) acur[r][rt] = anxt[r][rt];
        }
    }
}


// first column (within its 32-column tile) of register group g = registers 4g..4g+3
__device__ __forceinline__ int acc_col4(int g, int hh) { return 8 * g + 4 * hh; }
// row of register `reg` of a 32x32 result tile
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// ---------------------------------------------------------------------------------------------
// Point counts that only the device knows (gft_deform_backward_rows: the rows with an upstream gradient are counted by a
// kernel and nothing is read back, so the call can be captured in a HIP graph).  The launches are sized for the capacity
// (every buffer keeps the capacity's plane stride, `n_pad` of the argument structs); a kernel that is handed a plan takes
// its extents from it and its surplus workgroups return at once.  Written by k_deform_plan.
// ---------------------------------------------------------------------------------------------
struct DevPlan {
    int64_t n;              // points
    int64_t n_ext;          // points padded to DF_PAD: the rows the walks compute and the weight-gradient sums run over
    int tiles_per_split;    // dw_splits() of n_ext
    int splits;
};

// ---------------------------------------------------------------------------------------------
// forward walk
// ---------------------------------------------------------------------------------------------
struct FwdArgs {
    const DevPlan* plan;    // NULL: n is the host's
    int64_t n, n_pad, t_stride;
    int xm, tm;       // octaves of the xyz / t encodings (time_utils.py:64-65)
    const float* xyz; const float* t;
    const float* packed;
    float* emb;       // [n_pad][96] or null
    float* acts;      // [8][n_pad][256] or null
    uint32_t* signs;  // [8][n_pad][8] or null: bit c of word w = activation 32 w + c is positive
    float* d_xyz; float* d_sh;
    uint32_t gen;        // this forward call's number (fp16 walk: what it leaves in the range table when it cannot hold a value)
    uint32_t only_if;    // bf16 walk: 0 = run; else run only if the fp16 walk of call `only_if` gave up (see df_range_table)
};

constexpr int DF_BIAS_FLOATS = DF_D * DF_W + DF_HEAD;      // 2368
constexpr int DF_SIGN_WORDS = DF_W / 32;                   // ReLU sign bits of one point and layer: 8 words
constexpr size_t DF_FWD_LDS = ((size_t)32 * DF_NR_FWD * (DF_HS + DF_ES) + DF_BIAS_FLOATS) * 4;   // activations + encoding + all biases: 97536
constexpr size_t DF_BWD_LDS = (size_t)32 * DF_NR_BWD * DF_HS * 4;                               // 99840

template <bool SAVE>
__global__ __launch_bounds__(256) void k_deform_fwd(FwdArgs a)
{
    extern __shared__ float4 df_lds[];
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if ((int64_t)blockIdx.x * (32 * DF_NR_FWD) >= a.plan->n_ext) return;
        a.n = a.plan->n;
    }
    float* hA = reinterpret_cast<float*>(df_lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * (32 * DF_NR_FWD);
    const int n0 = wave * 64;
    const float4* wlane = reinterpret_cast<const float4*>(a.packed) + hh * DF_W + n0 + li;
    float4 wcur[2][2];
    load_w<2, DF_W>(wcur, wlane);
    // all biases (9.5 KB) go to LDS once: an epilogue then waits for an LDS read, not for L2
    float* eA = hA + (32 * DF_NR_FWD) * DF_HS;
    float* bL = eA + (32 * DF_NR_FWD) * DF_ES;
    for (int q = tid; q < DF_BIAS_FLOATS; q += 256) bL[q] = a.packed[DF_BIAS_BASE + q];

    // positional encoding (time_utils.py:24-53): [x, sin(2^f x), cos(2^f x)]_f for all dims, then t
    for (int pb = 0; pb < (32 * DF_NR_FWD); pb += 64) {
        const int pt = pb + (tid & 63), grp = tid >> 6;
        if (pt >= (32 * DF_NR_FWD)) break;
        const int64_t p = p0 + pt;
        float* e = eA + pt * DF_ES;
        if (grp < 3) {
            const float v = p < a.n ? a.xyz[3 * p + grp] : 0.f;
            e[grp] = v;
            for (int f = 0; f < a.xm; f++) {
                float sn, cs;
                sincosf(v * (float)(1 << f), &sn, &cs);
                e[3 + 6 * f + grp] = sn;
                e[6 + 6 * f + grp] = cs;
            }
        } else {
            const float v = p < a.n ? a.t[p * a.t_stride] : 0.f;
            const int t0 = 3 + 6 * a.xm;
            e[t0] = v;
            for (int f = 0; f < a.tm; f++) {
                float sn, cs;
                sincosf(v * (float)(1 << f), &sn, &cs);
                e[t0 + 1 + 2 * f] = sn;
                e[t0 + 2 + 2 * f] = cs;
            }
            for (int c = t0 + 1 + 2 * a.tm; c < DF_INK; c++) e[c] = 0.f;
        }
    }
    __syncthreads();
    if (SAVE) {
        for (int q = tid; q < (32 * DF_NR_FWD) * DF_EMB; q += 256) {
            const int row = q / DF_EMB, col = q - row * DF_EMB;
            a.emb[(p0 + row) * DF_EMB + col] = eA[row * DF_ES + col];
        }
    }

    const float* h_lane = hA + li * DF_HS + 4 * hh;
    const float* e_lane = eA + li * DF_ES + 4 * hh;
    f32x16 acc[DF_NR_FWD][2];
    for (int l = 0; l < DF_D; l++) {
        zero_acc(acc);
        if (l == 0) {
            stream_gemm<DF_NR_FWD, 2, DF_W>(acc, e_lane, DF_ES, DF_INK / 16, wlane, wcur, true);
        } else {
            // after layer 4 the encoding is concatenated in front (time_utils.py:112-113)
            if (l == 5) stream_gemm<DF_NR_FWD, 2, DF_W>(acc, e_lane, DF_ES, DF_INK / 16, wlane, wcur, true);
            stream_gemm<DF_NR_FWD, 2, DF_W>(acc, h_lane, DF_HS, DF_W / 16, wlane, wcur, l < 7);
        }
        // bias, ReLU -> next layer's A tile
        float4 bv[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                bv[ct][g] = *reinterpret_cast<const float4*>(bL + l * DF_W + n0 + 32 * ct + acc_col4(g, hh));
        __syncthreads();      // every wave is past its last read of this layer's input
#pragma unroll
        for (int rt = 0; rt < DF_NR_FWD; rt++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                uint32_t bits = 0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int row = 32 * rt + li, col = n0 + 32 * ct + acc_col4(g, hh);
                    float4 v;
                    v.x = fmaxf(acc[rt][ct][4 * g] + bv[ct][g].x, 0.f);
                    v.y = fmaxf(acc[rt][ct][4 * g + 1] + bv[ct][g].y, 0.f);
                    v.z = fmaxf(acc[rt][ct][4 * g + 2] + bv[ct][g].z, 0.f);
                    v.w = fmaxf(acc[rt][ct][4 * g + 3] + bv[ct][g].w, 0.f);
                    *reinterpret_cast<float4*>(hA + row * DF_HS + col) = v;
                    if (SAVE) {
                        *reinterpret_cast<float4*>(a.acts + ((int64_t)l * a.n_pad + p0 + row) * DF_W + col) = v;
                        bits |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                                << acc_col4(g, hh);
                    }
                }
                if (SAVE) {
                    // the other half wave holds the other 16 columns of this 32-column word
                    bits |= (uint32_t)__shfl_xor((int)bits, 32);
                    if (hh == 0) a.signs[((int64_t)l * a.n_pad + p0 + 32 * rt + li) * DF_SIGN_WORDS + 2 * wave + ct] = bits;
                }
            }
        __syncthreads();
    }
    // heads: 64 columns
    {
        const int ct = wave & 1;
        const float4* hl = reinterpret_cast<const float4*>(a.packed + DF_F_TOTAL - DF_F_HEAD_SZ) + hh * DF_HEAD + 32 * ct + li;
        float4 hw[2][1];
        load_w<1, DF_HEAD>(hw, hl);
        auto store_head = [&](const f32x16& t, int row0) {
            const int64_t p = p0 + row0 + li;
            if (p < a.n) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = 32 * ct + acc_col4(g, hh);
                    const float4 bq = *reinterpret_cast<const float4*>(bL + DF_D * DF_W + col);
                    const float4 v = make_float4(t[4 * g] + bq.x, t[4 * g + 1] + bq.y, t[4 * g + 2] + bq.z, t[4 * g + 3] + bq.w);
                    if (col < 48) {
                        *reinterpret_cast<float4*>(a.d_sh + p * 48 + col) = v;
                    } else if (col == 48) {
                        a.d_xyz[p * 3] = v.x;
                        a.d_xyz[p * 3 + 1] = v.y;
                        a.d_xyz[p * 3 + 2] = v.z;
                    }
                }
            }
        };
        if (DF_NR_FWD == 3) {
            // 3 row tiles x 2 column tiles over 4 waves: waves 0,1 take row tiles 0,1, waves 2,3 row tile 2
            if (wave < 2) {
                f32x16 hacc[2][1];
                zero_acc(hacc);
                stream_gemm<2, 1, DF_HEAD>(hacc, hA + li * DF_HS + 4 * hh, DF_HS, DF_W / 16, hl, hw, false);
                store_head(hacc[0][0], 0);
                store_head(hacc[1][0], 32);
            } else {
                f32x16 hacc[1][1];
                zero_acc(hacc);
                stream_gemm<1, 1, DF_HEAD>(hacc, hA + (64 + li) * DF_HS + 4 * hh, DF_HS, DF_W / 16, hl, hw, false);
                store_head(hacc[0][0], 64);
            }
        } else {
            // 2 x 2 tiles: one per wave
            const int r0 = 32 * (wave >> 1);
            f32x16 hacc[1][1];
            zero_acc(hacc);
            stream_gemm<1, 1, DF_HEAD>(hacc, hA + (r0 + li) * DF_HS + 4 * hh, DF_HS, DF_W / 16, hl, hw, false);
            store_head(hacc[0][0], r0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward walk on three bf16 planes: six bf16 MFMAs per product (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid;
// the three dropped terms are below 2^-23 of the product) accumulate in fp32 -- the error against float64 is
// that of an fp32 fma chain (profiles/experiments/bf16x3_gemm.hip: 5.4e-7 vs 5.5e-7), at up to 2.7 x the rate
// of v_mfma_f32_32x32x2_f32.  Activations live in LDS as three bf16 planes (the epilogue splits them),
// weights come as three bf16 planes from the packed copy.
// ---------------------------------------------------------------------------------------------
constexpr int DF_BH = 264;                 // bf16 per activation row of a plane (528 B: 16-byte reads of 16 rows hit 16 bank groups)
constexpr int DF_BE = 104;                 // bf16 per encoding row of a plane (208 B: 16-byte reads of 16 rows hit 16 bank groups)
constexpr size_t DF_BF_ACT_PLANE = (size_t)64 * DF_BH * 2, DF_BF_ENC_PLANE = (size_t)64 * DF_BE * 2;
constexpr size_t DF_FWD_BF_LDS = 3 * DF_BF_ACT_PLANE + 3 * DF_BF_ENC_PLANE + (size_t)DF_BIAS_FLOATS * 4;   // 144640

struct WSeg {                              // a weight segment as this lane sees it
    const uint4* p;                        // plane 0, chunk 0: [h][first column + lane]
    int plane;                             // uint4 between planes
};

__device__ __forceinline__ WSeg wseg(const __bf16* bf, int s, int lane_col, int hh)
{
    const DfSeg sg = df_seg(s);
    WSeg w;
    w.p = reinterpret_cast<const uint4*>(bf + 3 * sg.off) + hh * sg.ncol + lane_col;
    w.plane = sg.K * sg.ncol / 8;
    return w;
}

template <int NC, int NCOL>
__device__ __forceinline__ void load_wbf(uint4 (&w)[3][NC], const WSeg& sgp, int chunk)
{
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++) w[pl][ct] = sgp.p[(size_t)pl * sgp.plane + (size_t)(2 * chunk) * NCOL + 32 * ct];
}

__device__ __forceinline__ bf16x8 as_bf(const uint4& v)
{
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// acc[rt][ct] += (A[64 x 16 nchunks] W)^T; a_lane: this lane's row of plane 0 at k = 8 hh (bytes); w0, w1: chunks 0 and 1
// of `cur`, already loaded; `nxt`: the segment that follows (its chunks 0 and 1 are prefetched into w0, w1), or null.
// The weights come from L2 (every workgroup streams all of them): two chunks = 48 multiplies of lead.
template <int NR, int NC, int NCOL>
__device__ __forceinline__ void stream_gemm_bf(f32x16 (&acc)[NR][NC], const char* a_lane, int a_row_bytes, size_t a_plane_bytes,
                                               int nchunks, const WSeg& cur, const WSeg* nxt, uint4 (&w0)[3][NC],
                                               uint4 (&w1)[3][NC])
{
    uint4 acur[3][NR];
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int rt = 0; rt < NR; rt++)
            acur[pl][rt] = *reinterpret_cast<const uint4*>(a_lane + pl * a_plane_bytes + (size_t)rt * 32 * a_row_bytes);
    for (int c = 0; c < nchunks; c++) {
        const bool last = c + 1 >= nchunks;
        uint4 w2[3][NC], anxt[3][NR];
        // (unconditional loads from valid addresses: conditionally filled arrays end up in scratch)
        if (c + 2 < nchunks) load_wbf<NC, NCOL>(w2, cur, c + 2);
        else load_wbf<NC, NCOL>(w2, nxt ? *nxt : cur, c + 2 - nchunks);
        const int cn = last ? c : c + 1;
#pragma unroll
        for (int pl = 0; pl < 3; pl++)
#pragma unroll
            for (int rt = 0; rt < NR; rt++)
                anxt[pl][rt] = *reinterpret_cast<const uint4*>(a_lane + pl * a_plane_bytes + (size_t)rt * 32 * a_row_bytes + cn * 32);
        // weights are the MFMA's A operand, activations its B operand (point on the lane, see stream_gemm)
#pragma unroll
        for (int term = 0; term < 6; term++) {
            const int pw = term == 0 ? 0 : term == 1 ? 1 : term == 2 ? 0 : term == 3 ? 2 : term == 4 ? 0 : 1;
            const int pa = term == 0 ? 0 : term == 1 ? 0 : term == 2 ? 1 : term == 3 ? 0 : term == 4 ? 2 : 1;
#pragma unroll
            for (int rt = 0; rt < NR; rt++)
#pragma unroll
                for (int ct = 0; ct < NC; ct++)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(w0[pw][ct]), as_bf(acur[pa][rt]), acc[rt][ct], 0, 0, 0);
        }
        // the loads of the chunks to come go between the multiplies, one per gap (issued in a run in front of them
        // they hold the wave for their issue time with the matrix pipe idle)
#pragma unroll
        for (int i = 0; i < 3 * NC + 3 * NR; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
            if (i < 3 * NC) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                   // VMEM read
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                              // DS read
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
#pragma unroll
            for (int ct = 0; ct < NC; ct++) { w0[pl][ct] = w1[pl][ct]; w1[pl][ct] = w2[pl][ct]; }
#pragma unroll
            for (int rt = 0; rt < NR; rt++) acur[pl][rt] = anxt[pl][rt];
        }
    }
}

// four consecutive fp32 values of one point -> 4 bf16 in each of the three planes (split3 of each value; two values
// per v_cvt_pk_bf16_f32, which leaves them packed as the planes store them)
__device__ __forceinline__ void store_split4(char* plane0, size_t plane_bytes, size_t byte_off, const float4& v)
{
    uint2 ph, pm, pl;
    ph.x = cvt_pk_bf16(v.x, v.y);
    ph.y = cvt_pk_bf16(v.z, v.w);
    const float r0 = v.x - __uint_as_float(ph.x << 16), r1 = v.y - __uint_as_float(ph.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(ph.y << 16), r3 = v.w - __uint_as_float(ph.y & 0xffff0000u);
    pm.x = cvt_pk_bf16(r0, r1);
    pm.y = cvt_pk_bf16(r2, r3);
    pl.x = cvt_pk_bf16(r0 - __uint_as_float(pm.x << 16), r1 - __uint_as_float(pm.x & 0xffff0000u));
    pl.y = cvt_pk_bf16(r2 - __uint_as_float(pm.y << 16), r3 - __uint_as_float(pm.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(plane0 + byte_off) = ph;
    *reinterpret_cast<uint2*>(plane0 + plane_bytes + byte_off) = pm;
    *reinterpret_cast<uint2*>(plane0 + 2 * plane_bytes + byte_off) = pl;
}

constexpr int64_t DF_H_FLOATS = DF_BF_ELEMS;                   // both streams (round 6: the backward walk too): 2 planes x 2 bytes per weight
constexpr int64_t DF_FLAG_FLOATS = 4;                          // behind it: word 0 != 0 = a weight is outside the fp16 planes' range
constexpr int64_t DF_FLAG_OFF = DF_PACKED_FLOATS + DF_BF_FLOATS + DF_H_FLOATS;
constexpr float DF_H_MAX = 65504.f;                            // largest fp16

// The fp16 planes hold |weight| < 64 and |activation| < 4094 (fp16's 65504 over the plane scales).  The reference computes
// in fp32, so a network outside that range must still come out right: the pack kernel flags such weights in the packed
// buffer, the fp16 walk tracks the largest scaled activation it splits and, when one does not fit, writes its call's
// number into this table; gft_deform_forward launches the bf16 walk (fp32 range) right behind it with that number, and its
// workgroups return at once unless the flag or the table says the fp16 walk's results cannot be used -- then they
// overwrite all of them.  A table of 64 entries by call number keeps concurrent calls on different streams apart.
__device__ uint32_t df_range_table[64];

__device__ __forceinline__ bool fp16_walk_gave_up(const float* packed, uint32_t gen)
{
    const uint32_t wflag = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(packed + DF_FLAG_OFF));
    const uint32_t aflag = __hip_atomic_load(&df_range_table[gen & 63], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return wflag != 0 || aflag == gen;
}

// DF_FWD_WAVES waves per workgroup: 4 (a wave owns 64 output columns, one wave per SIMD) or 8 (32 columns, two waves per
// SIMD: the other wave multiplies while one waits for its weights or its LDS reads; 256 registers per wave).  Measured at
// 300 k points: 2.15-2.2 ms with 4, 1.96-2.02 ms with 8.
constexpr int DF_FWD_WAVES = 8;
constexpr int DF_FWD_NC = 8 / DF_FWD_WAVES;            // 32-column tiles per wave
template <bool SAVE>
__global__ __launch_bounds__(64 * DF_FWD_WAVES) void k_deform_fwd_bf(FwdArgs a)
{
    extern __shared__ float4 df_lds[];
    if (a.only_if != 0u && !fp16_walk_gave_up(a.packed, a.only_if)) return;
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if ((int64_t)blockIdx.x * 64 >= a.plan->n_ext) return;
        a.n = a.plan->n;
    }
    char* hP = reinterpret_cast<char*>(df_lds);                  // activation planes [3][64][264] bf16
    char* eP = hP + 3 * DF_BF_ACT_PLANE;                         // encoding planes   [3][64][88]  bf16
    float* bL = reinterpret_cast<float*>(eP + 3 * DF_BF_ENC_PLANE);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    constexpr int NC = DF_FWD_NC, NT = 64 * DF_FWD_WAVES;
    const int n0 = wave * 32 * NC;
    const __bf16* bf = reinterpret_cast<const __bf16*>(a.packed + DF_PACKED_FLOATS);
    WSeg seg = wseg(bf, 0, n0 + li, hh);
    uint4 wcur[3][NC], wnx1[3][NC];
    load_wbf<NC, DF_W>(wcur, seg, 0);
    load_wbf<NC, DF_W>(wnx1, seg, 1);
    for (int q = tid; q < DF_BIAS_FLOATS; q += NT) bL[q] = a.packed[DF_BIAS_BASE + q];

    // positional encoding (time_utils.py:24-53), split into the three planes; the fp32 values are saved for the
    // weight-gradient GEMMs
    {
        const int pt = tid & 63, grp = tid >> 6;
        const int64_t p = p0 + pt;
        auto put = [&](int col, float v) {
            __bf16 h, m, l;
            split3(v, h, m, l);
            __bf16* e = reinterpret_cast<__bf16*>(eP) + pt * DF_BE + col;
            e[0] = h;
            e[DF_BF_ENC_PLANE / 2] = m;
            e[DF_BF_ENC_PLANE] = l;
            if (SAVE) a.emb[p * DF_EMB + col] = v;
        };
        if (grp > 3) {
        } else if (grp < 3) {
            const float v = p < a.n ? a.xyz[3 * p + grp] : 0.f;
            put(grp, v);
            for (int f = 0; f < a.xm; f++) {
                float sn, cs;
                sincosf(v * (float)(1 << f), &sn, &cs);
                put(3 + 6 * f + grp, sn);
                put(6 + 6 * f + grp, cs);
            }
        } else {
            const float v = p < a.n ? a.t[p * a.t_stride] : 0.f;
            const int t0 = 3 + 6 * a.xm;
            put(t0, v);
            for (int f = 0; f < a.tm; f++) {
                float sn, cs;
                sincosf(v * (float)(1 << f), &sn, &cs);
                put(t0 + 1 + 2 * f, sn);
                put(t0 + 2 + 2 * f, cs);
            }
            for (int c = t0 + 1 + 2 * a.tm; c < DF_INK; c++) put(c, 0.f);
        }
    }
    __syncthreads();

    const char* h_lane = hP + (size_t)li * DF_BH * 2 + 16 * hh;
    const char* e_lane = eP + (size_t)li * DF_BE * 2 + 16 * hh;
    f32x16 acc[2][NC];
    for (int l = 0; l < DF_D; l++) {
        zero_acc(acc);
        if (l == 0) {
            const WSeg nx = wseg(bf, 1, n0 + li, hh);
            stream_gemm_bf<2, NC, DF_W>(acc, e_lane, DF_BE * 2, DF_BF_ENC_PLANE, DF_INK / 16, seg, &nx, wcur, wnx1);
            seg = nx;
        } else {
            if (l == 5) {
                // after layer 4 the encoding is concatenated in front (time_utils.py:112-113)
                const WSeg nx = wseg(bf, 6, n0 + li, hh);
                stream_gemm_bf<2, NC, DF_W>(acc, e_lane, DF_BE * 2, DF_BF_ENC_PLANE, DF_INK / 16, seg, &nx, wcur, wnx1);
                seg = nx;
            }
            // segments: layers 1..4 -> 1..4, encoding rows of 5 -> 5, hidden rows of 5, 6, 7 -> 6, 7, 8
            const int s_next = l < 4 ? l + 1 : l == 4 ? 5 : l < 7 ? l + 2 : -1;
            if (s_next >= 0) {
                const WSeg nx = wseg(bf, s_next, n0 + li, hh);
                stream_gemm_bf<2, NC, DF_W>(acc, h_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, seg, &nx, wcur, wnx1);
                seg = nx;
            } else {
                stream_gemm_bf<2, NC, DF_W>(acc, h_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, seg, nullptr, wcur, wnx1);
            }
        }
        float4 bv[NC][4];
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                bv[ct][g] = *reinterpret_cast<const float4*>(bL + l * DF_W + n0 + 32 * ct + acc_col4(g, hh));
        __syncthreads();      // every wave is past its last read of this layer's input
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
                uint32_t bits = 0;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int row = 32 * rt + li, col = n0 + 32 * ct + acc_col4(g, hh);
                    float4 v;
                    // (ReLU that keeps a NaN, as torch.relu does: fmaxf would turn it into 0)
                    auto relu = [](float x) { return x < 0.f ? 0.f : x; };
                    v.x = relu(acc[rt][ct][4 * g] + bv[ct][g].x);
                    v.y = relu(acc[rt][ct][4 * g + 1] + bv[ct][g].y);
                    v.z = relu(acc[rt][ct][4 * g + 2] + bv[ct][g].z);
                    v.w = relu(acc[rt][ct][4 * g + 3] + bv[ct][g].w);
                    store_split4(hP, DF_BF_ACT_PLANE, ((size_t)row * DF_BH + col) * 2, v);
                    if (SAVE) {
                        *reinterpret_cast<float4*>(a.acts + ((int64_t)l * a.n_pad + p0 + row) * DF_W + col) = v;
                        bits |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u))
                                << acc_col4(g, hh);
                    }
                }
                if (SAVE) {
                    bits |= (uint32_t)__shfl_xor((int)bits, 32);
                    if (hh == 0) a.signs[((int64_t)l * a.n_pad + p0 + 32 * rt + li) * DF_SIGN_WORDS + NC * wave + ct] = bits;
                }
            }
        __syncthreads();
    }
    // heads: 64 columns, one 32 x 32 tile per wave (of the first four)
    if (wave < 4) {
        const int ct = wave & 1, r0 = 32 * (wave >> 1);
        const WSeg hs = wseg(bf, 9, 32 * ct + li, hh);
        uint4 hw[3][1], hw1[3][1];
        load_wbf<1, DF_HEAD>(hw, hs, 0);
        load_wbf<1, DF_HEAD>(hw1, hs, 1);
        f32x16 hacc[1][1];
        zero_acc(hacc);
        stream_gemm_bf<1, 1, DF_HEAD>(hacc, hP + (size_t)(r0 + li) * DF_BH * 2 + 16 * hh, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, hs,
                                      nullptr, hw, hw1);
        const int64_t p = p0 + r0 + li;
        if (p < a.n) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 32 * ct + acc_col4(g, hh);
                const float4 bq = *reinterpret_cast<const float4*>(bL + DF_D * DF_W + col);
                const float4 v = make_float4(hacc[0][0][4 * g] + bq.x, hacc[0][0][4 * g + 1] + bq.y, hacc[0][0][4 * g + 2] + bq.z,
                                             hacc[0][0][4 * g + 3] + bq.w);
                if (col < 48) {
                    *reinterpret_cast<float4*>(a.d_sh + p * 48 + col) = v;
                } else if (col == 48) {
                    a.d_xyz[p * 3] = v.x;
                    a.d_xyz[p * 3 + 1] = v.y;
                    a.d_xyz[p * 3 + 2] = v.z;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward walk on TWO fp16 planes (round 4).  The bf16 split needs three planes and six multiplies per product
// because a bf16 carries 8 mantissa bits.  An fp16 carries 11: x = hi + lo with hi = fp16(x), lo = fp16(x - hi) holds 22
// bits, and a product is hi*hi + hi*lo + lo*hi -- THREE v_mfma_f32_32x32x16_f16 into one fp32 accumulator (the dropped
// lo*lo is below 2^-22 of the product).  fp16 has 5 exponent bits, so both operands are moved into its range first:
// weights are stored times 2^10, activations times 2^4, and the accumulator is scaled by 2^-14 once per tile.  With that
// a weight of magnitude 1e-4 .. 64 and an activation of 8e-3 .. 4094 have a NORMAL lo part (the full 22 bits); smaller
// ones have a subnormal lo, which still resolves 6e-8 of the scaled value (4e-9 of an activation, 6e-11 of a weight:
// absolute errors far below the fp32 rounding of the sums they enter) -- the matrix pipe honours fp16 subnormals
// (profiles/experiments/mfma_f16_subnormal.hip).  Measured against float64 (numpy model of the arithmetic, K = 256):
// 3-4e-7 of the max-norm, where a plain fp32 GEMM has 4.6e-7; the parity tests and their tolerances did not change.  Half
// the multiplies of the bf16 walk, two thirds of its operand bytes from L2 and LDS, a third of its accumulators.  The
// price is the range: a weight beyond 64 or an activation beyond 4094 becomes inf (and shows as inf / NaN in the
// outputs).  GFT_DEFORM_FP16X2=0 keeps the bf16 walk, which has the fp32 range.
// ---------------------------------------------------------------------------------------------
#ifndef DF_ABL
#define DF_ABL 0            // timing ablations of the fp16 forward (profiles/deform_ablate.sh): never set in the product build
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
constexpr float DF_H_WSCALE = 1024.0f, DF_H_ASCALE = 16.0f, DF_H_OUT = 1.0f / (1024.0f * 16.0f);
constexpr size_t DF_FWD_H_LDS = 2 * DF_BF_ACT_PLANE + 2 * DF_BF_ENC_PLANE + (size_t)DF_BIAS_FLOATS * 4;   // 99584
constexpr size_t DF_FWD_H_SAVE_LDS = 2 * DF_BF_ACT_PLANE + (size_t)DF_BIAS_FLOATS * 4;                    // 72960: two per CU
constexpr int DF_FWD_H_SAVE_WAVES = 4;

__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b)
{
    const f32x2_t v = {a, b};
    const f16x2_t r = __builtin_convertvector(v, f16x2_t);
    uint32_t u;
    __builtin_memcpy(&u, &r, 4);
    return u;
}
__device__ __forceinline__ float f16_lo(uint32_t u) { f16x2_t h; __builtin_memcpy(&h, &u, 4); return (float)h.x; }
__device__ __forceinline__ float f16_hi(uint32_t u) { f16x2_t h; __builtin_memcpy(&h, &u, 4); return (float)h.y; }

// fp32 packed streams ([k/4][n][4] per segment) -> fp16 planes ([plane][k/8][n][8] per segment)
__global__ __launch_bounds__(256) void k_deform_pack_h(const float* __restrict__ packed, _Float16* __restrict__ out, uint32_t* __restrict__ flag)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= DF_BF_ELEMS) return;
    int sgi = 0;
    for (int q = 1; q < DF_NSEG; q++)
        if (e >= df_seg(q).off) sgi = q;
    const DfSeg sg = df_seg(sgi);
    const int64_t r = e - sg.off;
    const int kq = (int)(r / (sg.ncol * 4)), n = (int)((r >> 2) % sg.ncol), k = 4 * kq + (int)(r & 3);
    const float x = packed[e] * DF_H_WSCALE;
    if (!(fabsf(x) <= DF_H_MAX)) flag[0] = 1u;          // (NaN too)
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    const int64_t plane = (int64_t)sg.K * sg.ncol;
    const int64_t o = 2 * sg.off + ((int64_t)(k >> 3) * sg.ncol + n) * 8 + (k & 7);
    out[o] = hi;
    out[o + plane] = lo;
}

// Memory operands of this kernel go through buffer instructions: a resource in four scalar registers, a wave-uniform
// byte offset in a fifth and this lane's 32-bit offset in ONE vector register (`buffer_load_dwordx4 v, v_off, s[rsrc],
// s_off offen offset:imm`).  With flat pointers every (plane, tile) of the weights, of the saved activations and of the
// encoding kept a 64-bit vector address alive across the walk -- dozens of registers that the accumulators and the
// prefetched operands need.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, const float4& v)
{
    u32x4_t u;
    __builtin_memcpy(&u, &v, 16);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)voff, (int)soff, 0);
}

// a weight segment of the fp16 stream
struct WSegH {
    uint32_t off;                          // bytes from the start of the stream to plane 0, chunk 0 (wave-uniform)
    uint32_t plane;                        // bytes between planes (wave-uniform)
    uint32_t lane;                         // bytes: [h][first column + lane]
};

__device__ __forceinline__ WSegH wseg_h(int s, int lane_col, int hh)
{
    const DfSeg sg = df_seg(s);
    WSegH w;
    w.off = (uint32_t)sg.off * 4u;
    w.plane = (uint32_t)(sg.K * sg.ncol) * 2u;
    w.lane = (uint32_t)(hh * sg.ncol + lane_col) * 16u;
    return w;
}

template <int NC, int NCOL>
__device__ __forceinline__ void load_wh(uint4 (&w)[2][NC], __amdgpu_buffer_rsrc_t rw, const WSegH& sgp, int chunk)
{
#pragma unroll
    for (int pl = 0; pl < 2; pl++)
#pragma unroll
        for (int ct = 0; ct < NC; ct++)
            w[pl][ct] = buf_load16(rw, sgp.lane + 512u * ct, sgp.off + (uint32_t)pl * sgp.plane + (uint32_t)((DF_ABL & 2) ? (chunk & 1) : chunk) * (2 * NCOL * 16));
}

__device__ __forceinline__ f16x8 as_h(const uint4& v)
{
    f16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// acc += hi * hi + hi * lo + lo * hi over `nchunks` 16-k chunks (see stream_gemm_bf for the operand roles).  With
// three multiplies per tile and chunk instead of six, a prefetch distance of two chunks is half the cycles it was: the
// weights of a chunk (from L2) are asked for THREE chunks ahead (w0 = current, w1, w2 in flight, the fourth set loaded
// here), the activations (from LDS) one chunk ahead; the loads go between the multiplies.
template <int NR, int NC, int NCOL>
__device__ __forceinline__ void stream_gemm_h(f32x16 (&acc)[NR][NC], const char* a_lane, int a_row_bytes,
                                              size_t a_plane_bytes, int nchunks, __amdgpu_buffer_rsrc_t rw, const WSegH& cur, const WSegH* nxt, uint4 (&w0)[2][NC],
                                              uint4 (&w1)[2][NC], uint4 (&w2)[2][NC])
{
    uint4 acur[2][NR];
#pragma unroll
    for (int pl = 0; pl < 2; pl++)
#pragma unroll
        for (int rt = 0; rt < NR; rt++)
            acur[pl][rt] = *reinterpret_cast<const uint4*>(a_lane + pl * a_plane_bytes + (size_t)rt * 32 * a_row_bytes);
    for (int c = 0; c < nchunks; c++) {
        const bool last = c + 1 >= nchunks;
        uint4 w3[2][NC], anxt[2][NR];
        // (unconditional loads from valid addresses: conditionally filled arrays end up in scratch)
        if (c + 3 < nchunks) load_wh<NC, NCOL>(w3, rw, cur, c + 3);
        else load_wh<NC, NCOL>(w3, rw, nxt ? *nxt : cur, c + 3 - nchunks);
        const int cn = last ? c : c + 1;
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int rt = 0; rt < NR; rt++)
                anxt[pl][rt] = *reinterpret_cast<const uint4*>(a_lane + pl * a_plane_bytes + (size_t)rt * 32 * a_row_bytes + cn * 32);
        // term-major: consecutive multiplies go to different accumulators
        // term-major: consecutive multiplies go to different accumulators
#pragma unroll
        for (int term = 0; term < ((DF_ABL & 8) ? 1 : 3); term++)
#pragma unroll
            for (int rt = 0; rt < NR; rt++)
#pragma unroll
                for (int ct = 0; ct < NC; ct++)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(w0[term == 2 ? 1 : 0][ct]), as_h(acur[term == 1 ? 1 : 0][rt]),
                                                                         acc[rt][ct], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2 * NC + 2 * NR; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
            if (i < 2 * NC) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                   // VMEM read
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                              // DS read
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
#pragma unroll
            for (int ct = 0; ct < NC; ct++) { w0[pl][ct] = w1[pl][ct]; w1[pl][ct] = w2[pl][ct]; w2[pl][ct] = w3[pl][ct]; }
#pragma unroll
            for (int rt = 0; rt < NR; rt++) acur[pl][rt] = anxt[pl][rt];
        }
    }
}

// sin and cos of a moderate argument (the encoding's 2^f x with x in [0, 1]: below 2^15 here, beyond that the library
// routine): three-constant Cody-Waite reduction by pi/2 with fused multiply-adds, then the Cephes single-precision
// polynomials on [-pi/4, pi/4] -- about 1 ulp, a third of the instructions of the library's sincosf, which also carries
// the large-argument reduction.  The encoding phase was a tenth of the kernel.
__device__ __forceinline__ void sincos_enc(float a, float& sn, float& cs)
{
    if (!(fabsf(a) < 32768.f)) {
        sincosf(a, &sn, &cs);
        return;
    }
    const float k = rintf(a * 0.636619772367581343f);
    float r = fmaf(-k, 1.57079625129699707031f, a);
    r = fmaf(-k, 7.54978941586159635335e-08f, r);
    r = fmaf(-k, 5.39030252995776476554e-15f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k;
    const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s0 : s0;
    cs = ((q + 1) & 2) ? -c0 : c0;
}

// x - (float)h for the low / high half of a packed fp16 pair in ONE instruction (the mixed-precision fma converts the half)
__device__ __forceinline__ float sub_f16_lo(float x, uint32_t h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
__device__ __forceinline__ float sub_f16_hi(float x, uint32_t h)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
    return r;
}
// 1 if the fp32 value is positive, else 0 (one v_med3_i32: -0 and +0 are not positive integers)
__device__ __forceinline__ uint32_t positive_bit(float v)
{
    uint32_t b;
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(b) : "v"(v));      // (written as min / max the compiler makes it a compare and a select)
    return b;
}

__device__ __forceinline__ float4 as_f4(const uint4& u)
{
    float4 f;
    __builtin_memcpy(&f, &u, 16);
    return f;
}

// eight consecutive fp32 values -> the two fp16 operand fragments (hi, lo of the scaled value), as store_split4_h writes them
__device__ __forceinline__ void split8_h(float4 u, float4 v, uint4& hi, uint4& lo)
{
    u.x *= DF_H_ASCALE; u.y *= DF_H_ASCALE; u.z *= DF_H_ASCALE; u.w *= DF_H_ASCALE;
    v.x *= DF_H_ASCALE; v.y *= DF_H_ASCALE; v.z *= DF_H_ASCALE; v.w *= DF_H_ASCALE;
    hi.x = cvt_pk_f16(u.x, u.y);
    hi.y = cvt_pk_f16(u.z, u.w);
    hi.z = cvt_pk_f16(v.x, v.y);
    hi.w = cvt_pk_f16(v.z, v.w);
    lo.x = cvt_pk_f16(sub_f16_lo(u.x, hi.x), sub_f16_hi(u.y, hi.x));
    lo.y = cvt_pk_f16(sub_f16_lo(u.z, hi.y), sub_f16_hi(u.w, hi.y));
    lo.z = cvt_pk_f16(sub_f16_lo(v.x, hi.z), sub_f16_hi(v.y, hi.z));
    lo.w = cvt_pk_f16(sub_f16_lo(v.z, hi.w), sub_f16_hi(v.w, hi.w));
}

// stream_gemm_h with the left operand read as fp32 rows of global memory (the saved encoding, written by this workgroup
// before its first barrier) and split into the two planes in registers: the skip connection of layer 5 without the encoding
// resident in LDS.  `re` = the workgroup's 64 rows, `e_lane` = byte offset of this lane's row of the first
// 32-row tile at column 8 * (lane / 32).
template <int NC, int NCOL>
__device__ __forceinline__ void stream_gemm_h_emb(f32x16 (&acc)[2][NC], __amdgpu_buffer_rsrc_t re, uint32_t e_lane, int nchunks,
                                                  __amdgpu_buffer_rsrc_t rw, const WSegH& cur, const WSegH* nxt, uint4 (&w0)[2][NC], uint4 (&w1)[2][NC],
                                                  uint4 (&w2)[2][NC])
{
    uint4 raw[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; rt++) {
        raw[rt][0] = buf_load16(re, e_lane, rt * (32 * DF_EMB * 4));
        raw[rt][1] = buf_load16(re, e_lane + 16, rt * (32 * DF_EMB * 4));
    }
    for (int c = 0; c < nchunks; c++) {
        const bool last = c + 1 >= nchunks;
        uint4 acur[2][2], w3[2][NC];
#pragma unroll
        for (int rt = 0; rt < 2; rt++) split8_h(as_f4(raw[rt][0]), as_f4(raw[rt][1]), acur[0][rt], acur[1][rt]);
        __builtin_amdgcn_sched_barrier(0);
        if (c + 3 < nchunks) load_wh<NC, NCOL>(w3, rw, cur, c + 3);
        else load_wh<NC, NCOL>(w3, rw, nxt ? *nxt : cur, c + 3 - nchunks);
        const int cn = last ? c : c + 1;
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
            raw[rt][0] = buf_load16(re, e_lane, rt * (32 * DF_EMB * 4) + cn * 64);
            raw[rt][1] = buf_load16(re, e_lane + 16, rt * (32 * DF_EMB * 4) + cn * 64);
        }
#pragma unroll
        for (int term = 0; term < 3; term++)
#pragma unroll
            for (int rt = 0; rt < 2; rt++)
#pragma unroll
                for (int ct = 0; ct < NC; ct++)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(w0[term == 2 ? 1 : 0][ct]), as_h(acur[term == 1 ? 1 : 0][rt]),
                                                                         acc[rt][ct], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2 * NC + 4; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                   // VMEM read
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) { w0[pl][ct] = w1[pl][ct]; w1[pl][ct] = w2[pl][ct]; w2[pl][ct] = w3[pl][ct]; }
    }
}

// four consecutive fp32 values of one point, ALREADY times the activation scale -> 4 fp16 in each of the two planes (hi, lo)
__device__ __forceinline__ void store_split4_h(char* plane0, size_t plane_bytes, size_t byte_off, const float4& v)
{
    uint2 ph, pl;
    ph.x = cvt_pk_f16(v.x, v.y);
    ph.y = cvt_pk_f16(v.z, v.w);
    pl.x = cvt_pk_f16(sub_f16_lo(v.x, ph.x), sub_f16_hi(v.y, ph.x));
    pl.y = cvt_pk_f16(sub_f16_lo(v.z, ph.y), sub_f16_hi(v.w, ph.y));
    *reinterpret_cast<uint2*>(plane0 + byte_off) = ph;
    *reinterpret_cast<uint2*>(plane0 + plane_bytes + byte_off) = pl;
}

// Two shapes of the same walk.  Inference (SAVE = false): 8 waves of 32 columns, the encoding resident in LDS for the skip
// connection, one workgroup per CU.  Training (SAVE = true): the encoding is in global memory anyway (saved for the weight
// gradients), so it is built in the activation rows for layer 0 and read back from there for layer 5; without its planes the
// workgroup needs 73 KB of LDS and TWO fit a CU, as 4 waves of 64 columns each (the register budget of 2 waves per SIMD
// stays): one workgroup's epilogue (bias, ReLU, split, stores: no matrix work, all its waves between the same two
// barriers) now overlaps the other's multiplies, and a wave reads half the LDS bytes per multiply.
template <bool SAVE, int NW>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_deform_fwd_h(FwdArgs a)
{
    extern __shared__ float4 df_lds[];
    // a weight outside the planes' range: the bf16 walk launched behind this one does the work
    if (__builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.packed + DF_FLAG_OFF)) != 0u) return;
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if ((int64_t)blockIdx.x * 64 >= a.plan->n_ext) return;
        a.n = a.plan->n;
    }
    float top = 0.f;                   // largest scaled value this lane has split into the planes
    bool bad = false;                  // an encoded input that is NaN or beyond the planes' range
    constexpr bool ENC_LDS = !SAVE;
    char* hP = reinterpret_cast<char*>(df_lds);                  // activation planes [2][64][264] fp16
    char* eP = ENC_LDS ? hP + 2 * DF_BF_ACT_PLANE : hP;          // encoding planes   [2][64][104] fp16, or in the activation rows
    float* bL = reinterpret_cast<float*>(hP + 2 * DF_BF_ACT_PLANE + (ENC_LDS ? 2 * DF_BF_ENC_PLANE : 0));
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    constexpr int NC = 8 / NW, NT = 64 * NW;
    constexpr int E_ROW = ENC_LDS ? DF_BE : DF_BH;               // fp16 per encoding row
    constexpr size_t E_PLANE = ENC_LDS ? DF_BF_ENC_PLANE : DF_BF_ACT_PLANE;
    const int n0 = wave * 32 * NC;
    // (DF_ABL & 16: phase time stamps of every wave into the workgroup's d_sh rows instead of the results)
    int stamp_i = 0;
    auto stamp = [&]() {
        if (DF_ABL & 16) {
            const uint64_t tt = __builtin_amdgcn_s_memtime();
            if (lane == 0) reinterpret_cast<uint64_t*>(a.d_sh + p0 * 48)[wave * 48 + stamp_i] = tt;
            stamp_i++;
        }
    };
    stamp();
    const __amdgpu_buffer_rsrc_t rw = buf_rsrc(a.packed + DF_PACKED_FLOATS + DF_BF_FLOATS, (uint32_t)DF_H_FLOATS * 4u);
    WSegH seg = wseg_h(0, n0 + li, hh);
    uint4 wcur[2][NC], wnx1[2][NC], wnx2[2][NC];
    load_wh<NC, DF_W>(wcur, rw, seg, 0);
    load_wh<NC, DF_W>(wnx1, rw, seg, 1);
    load_wh<NC, DF_W>(wnx2, rw, seg, 2);
    // this thread's input of the encoding (asked for before anything waits) and the biases (all loads, then all stores)
    const float enc_in = (p0 + (tid & 63)) < a.n ? (((tid >> 6) & 3) < 3 ? a.xyz[3 * (p0 + (tid & 63)) + ((tid >> 6) & 3)] : a.t[(p0 + (tid & 63)) * a.t_stride]) : 0.f;
    {
        constexpr int NB = (DF_BIAS_FLOATS + NT - 1) / NT;
        float bv[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) bv[q] = tid + q * NT < DF_BIAS_FLOATS ? a.packed[DF_BIAS_BASE + tid + q * NT] : 0.f;
#pragma unroll
        for (int q = 0; q < NB; q++)
            if (tid + q * NT < DF_BIAS_FLOATS) bL[tid + q * NT] = bv[q];
    }

    // positional encoding (time_utils.py:24-53), split into the two planes; the fp32 values are saved for the
    // weight-gradient GEMMs
    {
        const int pt = tid & 63, grp = tid >> 6;
        const int64_t p = p0 + pt;
        auto put = [&](int col, float v) {
            const float vs = v * DF_H_ASCALE;
            // (fmaxf drops a NaN operand: an input that is NaN -- or infinite: its sine is NaN -- is caught by the comparison,
            // which is false for it.  Activations need no such test: with finite inputs and weights a NaN can only come
            // from values beyond the fp32 range, and those pass `top` on their way)
            bad |= !(fabsf(vs) <= DF_H_MAX);
            top = fmaxf(top, fabsf(vs));
            const _Float16 h = (_Float16)vs;
            _Float16* e = reinterpret_cast<_Float16*>(eP) + pt * E_ROW + col;
            e[0] = h;
            e[E_PLANE / 2] = (_Float16)(vs - (float)h);
            if (SAVE) a.emb[p * DF_EMB + col] = v;
        };
        // a 64-thread group per input (x, y, z, t); with 8 waves two groups share an input's octaves
        constexpr int NH = NW / 4;
        const int inp = grp & 3, half = grp >> 2;
        if (inp < 3) {
            const float v = enc_in;
            if (half == 0) put(inp, v);
            for (int f = a.xm * half / NH; f < a.xm * (half + 1) / NH; f++) {
                float sn, cs;
                sincos_enc(v * (float)(1 << f), sn, cs);
                put(3 + 6 * f + inp, sn);
                put(6 + 6 * f + inp, cs);
            }
        } else {
            const float v = enc_in;
            const int t0 = 3 + 6 * a.xm;
            if (half == 0) put(t0, v);
            for (int f = a.tm * half / NH; f < a.tm * (half + 1) / NH; f++) {
                float sn, cs;
                sincos_enc(v * (float)(1 << f), sn, cs);
                put(t0 + 1 + 2 * f, sn);
                put(t0 + 2 + 2 * f, cs);
            }
            if (half == NH - 1)
                for (int c = t0 + 1 + 2 * a.tm; c < DF_INK; c++) put(c, 0.f);
        }
    }
    stamp();
    __syncthreads();
    stamp();

    const char* h_lane = hP + (size_t)li * DF_BH * 2 + 16 * hh;
    const char* e_lane = eP + (size_t)li * E_ROW * 2 + 16 * hh;
    f32x16 acc[2][NC];
    for (int l = 0; l < DF_D; l++) {
        // (the accumulators start at zero and the bias is added in the epilogue: started AT the scaled bias the walk took
        // 0.45 ms longer and its error against float64 grew five-fold -- measured, profiles/README)
        zero_acc(acc);
        if (l == 0) {
            const WSegH nx = wseg_h(1, n0 + li, hh);
            stream_gemm_h<2, NC, DF_W>(acc, e_lane, E_ROW * 2, E_PLANE, DF_INK / 16, rw, seg, &nx, wcur, wnx1, wnx2);
            seg = nx;
        } else {
            if (l == 5) {
                const WSegH nx = wseg_h(6, n0 + li, hh);
                if (ENC_LDS)
                    stream_gemm_h<2, NC, DF_W>(acc, e_lane, E_ROW * 2, E_PLANE, DF_INK / 16, rw, seg, &nx, wcur, wnx1, wnx2);
                else
                    stream_gemm_h_emb<NC, DF_W>(acc, buf_rsrc(a.emb + p0 * DF_EMB, 64u * DF_EMB * 4u), (uint32_t)(li * DF_EMB + 8 * hh) * 4u, DF_INK / 16, rw, seg, &nx, wcur, wnx1, wnx2);
                seg = nx;
            }
            const int s_next = l < 4 ? l + 1 : l == 4 ? 5 : l < 7 ? l + 2 : -1;
            if (s_next >= 0) {
                const WSegH nx = wseg_h(s_next, n0 + li, hh);
                stream_gemm_h<2, NC, DF_W>(acc, h_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, rw, seg, &nx, wcur, wnx1, wnx2);
                seg = nx;
            } else {
                stream_gemm_h<2, NC, DF_W>(acc, h_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, rw, seg, nullptr, wcur, wnx1, wnx2);
            }
        }
        stamp();
        __syncthreads();      // every wave is past its last read of this layer's input
        stamp();
        // Sixteen 4-value groups per lane.  Every address below is ONE lane register plus an immediate: the registers are
        // made opaque per layer so that the sixteen sums are not hoisted out of the layer loop as sixteen registers each
        // (they were, and were spilled).
        uint32_t st_lane = (uint32_t)(li * DF_BH + n0 + 4 * hh) * 2u;     // bytes into plane 0 of the activations
        uint32_t sv_lane = (uint32_t)(li * DF_W + n0 + 4 * hh) * 4u;      // bytes into the workgroup's saved rows
        asm volatile("" : "+v"(st_lane), "+v"(sv_lane));
        char* st = hP + st_lane;
        const __amdgpu_buffer_rsrc_t ra = buf_rsrc(SAVE ? a.acts + ((int64_t)l * a.n_pad + p0) * DF_W : nullptr, SAVE ? 64u * DF_W * 4u : 0u);
        const __amdgpu_buffer_rsrc_t rs =
            buf_rsrc(SAVE ? a.signs + ((int64_t)l * a.n_pad + p0) * DF_SIGN_WORDS : nullptr, SAVE ? 64u * DF_SIGN_WORDS * 4u : 0u);
        uint32_t b_lane = (uint32_t)(n0 + 4 * hh) * 4u;                   // bytes into the layer's biases
        asm volatile("" : "+v"(b_lane));
        const char* bq_l = reinterpret_cast<const char*>(bL + l * DF_W) + b_lane;
#pragma unroll
        for (int ct = 0; ct < NC; ct++) {
            f32x2_t bq[4][2];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const float4 q = *reinterpret_cast<const float4*>(bq_l + (32 * ct + 8 * g) * 4);
                bq[g][0] = f32x2_t{q.x, q.y};
                bq[g][1] = f32x2_t{q.z, q.w};
            }
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                uint32_t bits = 0;
#pragma unroll
                for (int g = 3; g >= 0; g--) {
                    const size_t lds_off = (size_t)(rt * 32 * DF_BH + 32 * ct + 8 * g) * 2;
                    if (DF_ABL & 1) {
                        uint2 ph;
                        ph.x = cvt_pk_f16(acc[rt][ct][4 * g], acc[rt][ct][4 * g + 1]);
                        ph.y = cvt_pk_f16(acc[rt][ct][4 * g + 2], acc[rt][ct][4 * g + 3]);
                        *reinterpret_cast<uint2*>(st + lds_off) = ph;
                        continue;
                    }
                    // activation = max(acc 2^-14 + bias, 0): packed fused multiply-adds, two values each
                    const f32x2_t a01 = __builtin_elementwise_fma(f32x2_t{acc[rt][ct][4 * g], acc[rt][ct][4 * g + 1]}, f32x2_t{DF_H_OUT, DF_H_OUT}, bq[g][0]);
                    const f32x2_t a23 = __builtin_elementwise_fma(f32x2_t{acc[rt][ct][4 * g + 2], acc[rt][ct][4 * g + 3]}, f32x2_t{DF_H_OUT, DF_H_OUT}, bq[g][1]);
                    const float4 v = make_float4(fmaxf(a01.x, 0.f), fmaxf(a01.y, 0.f), fmaxf(a23.x, 0.f), fmaxf(a23.y, 0.f));
                    const f32x2_t s01 = f32x2_t{v.x, v.y} * DF_H_ASCALE, s23 = f32x2_t{v.z, v.w} * DF_H_ASCALE;
                    top = fmaxf(fmaxf(top, fmaxf(s01.x, s01.y)), fmaxf(s23.x, s23.y));
                    store_split4_h(st, DF_BF_ACT_PLANE, lds_off, make_float4(s01.x, s01.y, s23.x, s23.y));
                    if (SAVE && !(DF_ABL & 4)) {
                        // (the row tile goes into the lane offset, not into the scalar offset: a 16-byte buffer store with a
                        // scalar-register offset was seen to pick up the NEXT values of its data registers in its last lanes
                        // on gfx950 -- the compiler pads that hazard only for stores without one)
                        buf_store16(ra, sv_lane + (uint32_t)(rt * 32 * DF_W + 32 * ct + 8 * g) * 4u, 0, v);
                        // sign bits, highest column first: bit 8 g + 4 hh + j of the tile's word
                        bits = (bits << 8) | (positive_bit(v.w) << 3) | (positive_bit(v.z) << 2) | (positive_bit(v.y) << 1) | positive_bit(v.x);
                    }
                }
                if (SAVE && !(DF_ABL & 5)) {
                    bits <<= 4 * hh;
                    bits |= (uint32_t)__shfl_xor((int)bits, 32);
                    if (hh == 0)
                        __builtin_amdgcn_raw_buffer_store_b32(bits, rs, (int)((uint32_t)(li * DF_SIGN_WORDS + NC * wave + ct) * 4u),
                                                              rt * (32 * DF_SIGN_WORDS * 4), 0);
                }
            }
        }
        stamp();
        __syncthreads();
        stamp();
    }
    // heads: 64 columns, one 32 x 32 tile per wave (of the first four)
    if (wave < 4) {
        const int ct = wave & 1, r0 = 32 * (wave >> 1);
        const WSegH hs = wseg_h(9, 32 * ct + li, hh);
        uint4 hw[2][1], hw1[2][1], hw2[2][1];
        load_wh<1, DF_HEAD>(hw, rw, hs, 0);
        load_wh<1, DF_HEAD>(hw1, rw, hs, 1);
        load_wh<1, DF_HEAD>(hw2, rw, hs, 2);
        f32x16 hacc[1][1];
        zero_acc(hacc);
        stream_gemm_h<1, 1, DF_HEAD>(hacc, hP + (size_t)(r0 + li) * DF_BH * 2 + 16 * hh, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16,
                                     rw, hs, nullptr, hw, hw1, hw2);
        const int64_t p = p0 + r0 + li;
        if (p < a.n && !(DF_ABL & 16)) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int col = 32 * ct + acc_col4(g, hh);
                const float4 bq = *reinterpret_cast<const float4*>(bL + DF_D * DF_W + col);
                const float4 v = make_float4(fmaf(hacc[0][0][4 * g], DF_H_OUT, bq.x), fmaf(hacc[0][0][4 * g + 1], DF_H_OUT, bq.y),
                                             fmaf(hacc[0][0][4 * g + 2], DF_H_OUT, bq.z), fmaf(hacc[0][0][4 * g + 3], DF_H_OUT, bq.w));
                if (col < 48) {
                    *reinterpret_cast<float4*>(a.d_sh + p * 48 + col) = v;
                } else if (col == 48) {
                    a.d_xyz[p * 3] = v.x;
                    a.d_xyz[p * 3 + 1] = v.y;
                    a.d_xyz[p * 3 + 2] = v.z;
                }
            }
        }
    }
    stamp();
    // (a NaN input counts as "does not fit" through `bad`; the fp32-range walk behind this one then propagates it as
    // torch.relu does)
    if (bad || !(top <= DF_H_MAX)) __hip_atomic_store(&df_range_table[a.gen & 63], a.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// backward walk: dz_l = dh_l * [act_l > 0], dh_{l-1} = dz_l W_l
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
    const DevPlan* plan;    // NULL: n is the host's
    int64_t n, n_pad;
    const float* packed;
    const uint32_t* signs;  // [8][n_pad][8]
    const float* g_dxyz; const float* g_dsh;
    float* dz;              // [8][n_pad][256]
    float* dzh;             // [n_pad][64]
    int only_if_wflag;      // k_deform_bwd_bf behind k_deform_bwd_h: run only if a weight is outside the fp16 planes' range
    float* rowmax;          // [8][n_pad]: k_deform_bwd_h leaves the largest |dz_l| of every point (layers 1..7) for k_deform_dw_h's scales
    uint32_t* xflag;        // cleared by k_deform_bwd_h; k_deform_dw_h sets it when an activation does not fit its fp16 planes
};

__global__ __launch_bounds__(256) void k_deform_bwd(BwdArgs a)
{
    extern __shared__ float4 df_lds[];
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if ((int64_t)blockIdx.x * (32 * DF_NR_BWD) >= a.plan->n_ext) return;
        a.n = a.plan->n;
    }
    float* gA = reinterpret_cast<float*>(df_lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * (32 * DF_NR_BWD);
    const int n0 = wave * 64;
    const float4* wlane = reinterpret_cast<const float4*>(a.packed + DF_B_BASE) + hh * DF_W + n0 + li;
    float4 wcur[2][2];
    load_w<2, DF_W>(wcur, wlane);

    // upstream gradients as the head's output tile: [d_sh (48) | d_xyz (3) | 0], in the first 64 columns
    for (int q = tid; q < (32 * DF_NR_BWD) * DF_HEAD; q += 256) {
        const int row = q >> 6, col = q & 63;
        const int64_t p = p0 + row;
        float v = 0.f;
        if (p < a.n) {
            if (col < 48) v = a.g_dsh ? a.g_dsh[p * 48 + col] : 0.f;
            else if (col < 51) v = a.g_dxyz ? a.g_dxyz[p * 3 + (col - 48)] : 0.f;
        }
        gA[row * DF_HS + col] = v;
        a.dzh[p * DF_HEAD + col] = v;
    }
    __syncthreads();

    const float* g_lane = gA + li * DF_HS + 4 * hh;
    f32x16 acc[DF_NR_BWD][2];
    // ReLU signs of the lane's point and the wave's 64 columns (2 words per row tile), fetched before the
    // multiply that produces the gradient they gate
    uint2 sg[DF_NR_BWD];
    auto load_signs = [&](int l) {
#pragma unroll
        for (int rt = 0; rt < DF_NR_BWD; rt++)
            sg[rt] = *reinterpret_cast<const uint2*>(a.signs + ((int64_t)l * a.n_pad + p0 + 32 * rt + li) * DF_SIGN_WORDS + 2 * wave);
    };
    load_signs(DF_D - 1);
    zero_acc(acc);
    stream_gemm<DF_NR_BWD, 2, DF_W>(acc, g_lane, DF_HS, DF_HEAD / 16, wlane, wcur, true);   // dh_7
    for (int l = DF_D - 1; l >= 0; l--) {
        __syncthreads();      // every wave is past its last read of the previous gradient tile
#pragma unroll
        for (int rt = 0; rt < DF_NR_BWD; rt++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int row = 32 * rt + li;
                const uint32_t word = ct == 0 ? sg[rt].x : sg[rt].y;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = n0 + 32 * ct + acc_col4(g, hh);
                    const uint32_t m = word >> acc_col4(g, hh);
                    float4 v;
                    v.x = (m & 1u) ? acc[rt][ct][4 * g] : 0.f;
                    v.y = (m & 2u) ? acc[rt][ct][4 * g + 1] : 0.f;
                    v.z = (m & 4u) ? acc[rt][ct][4 * g + 2] : 0.f;
                    v.w = (m & 8u) ? acc[rt][ct][4 * g + 3] : 0.f;
                    if (l > 0) *reinterpret_cast<float4*>(gA + row * DF_HS + col) = v;
                    *reinterpret_cast<float4*>(a.dz + ((int64_t)l * a.n_pad + p0 + row) * DF_W + col) = v;
                }
            }
        if (l == 0) break;
        __syncthreads();
        load_signs(l - 1);
        zero_acc(acc);
        stream_gemm<DF_NR_BWD, 2, DF_W>(acc, g_lane, DF_HS, DF_W / 16, wlane, wcur, l > 1);   // dh_{l-1}
    }
}

// backward walk on three bf16 planes (64 points per workgroup): as k_deform_bwd, with the gradient tile in LDS
// as hi / mid / lo planes and the weights from the bf16 copy of the backward stream
constexpr size_t DF_BWD_BF_LDS = 3 * DF_BF_ACT_PLANE;   // 101376

constexpr int DF_BWD_WAVES = 4;                        // as DF_FWD_WAVES; measured at 300 k points: 1.64 ms with 4, 1.76 ms with 8
constexpr int DF_BWD_NC = 8 / DF_BWD_WAVES;
__global__ __launch_bounds__(64 * DF_BWD_WAVES) void k_deform_bwd_bf(BwdArgs a)
{
    extern __shared__ float4 df_lds[];
    if (a.only_if_wflag && __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.packed + DF_FLAG_OFF)) == 0u) return;
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if ((int64_t)blockIdx.x * 64 >= a.plan->n_ext) return;
        a.n = a.plan->n;
    }
    char* gP = reinterpret_cast<char*>(df_lds);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    constexpr int NC = DF_BWD_NC, NT = 64 * DF_BWD_WAVES;
    const int n0 = wave * 32 * NC;
    const __bf16* bf = reinterpret_cast<const __bf16*>(a.packed + DF_PACKED_FLOATS);
    WSeg seg = wseg(bf, 10, n0 + li, hh);
    uint4 wcur[3][NC], wnx1[3][NC];
    load_wbf<NC, DF_W>(wcur, seg, 0);
    load_wbf<NC, DF_W>(wnx1, seg, 1);

    // upstream gradients as the head's output tile: [d_sh (48) | d_xyz (3) | 0], in the first 64 columns
    for (int q = tid; q < 64 * (DF_HEAD / 4); q += NT) {
        const int row = q >> 4, col = (q & 15) * 4;
        const int64_t p = p0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p < a.n) {
            if (col < 48) {
                if (a.g_dsh) v = *reinterpret_cast<const float4*>(a.g_dsh + p * 48 + col);
            } else if (col == 48 && a.g_dxyz) {
                v.x = a.g_dxyz[p * 3]; v.y = a.g_dxyz[p * 3 + 1]; v.z = a.g_dxyz[p * 3 + 2];
            }
        }
        store_split4(gP, DF_BF_ACT_PLANE, ((size_t)row * DF_BH + col) * 2, v);
        *reinterpret_cast<float4*>(a.dzh + p * DF_HEAD + col) = v;
    }
    __syncthreads();

    const char* g_lane = gP + (size_t)li * DF_BH * 2 + 16 * hh;
    f32x16 acc[2][NC];
    uint32_t sg[2][NC];
    auto load_signs = [&](int l) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++)
                sg[rt][ct] = a.signs[((int64_t)l * a.n_pad + p0 + 32 * rt + li) * DF_SIGN_WORDS + NC * wave + ct];
    };
    load_signs(DF_D - 1);
    zero_acc(acc);
    {
        const WSeg nx = wseg(bf, 11, n0 + li, hh);
        stream_gemm_bf<2, NC, DF_W>(acc, g_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_HEAD / 16, seg, &nx, wcur, wnx1);   // dh_7
        seg = nx;
    }
    for (int l = DF_D - 1; l >= 0; l--) {
        __syncthreads();      // every wave is past its last read of the previous gradient tile
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
                const int row = 32 * rt + li;
                const uint32_t word = sg[rt][ct];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = n0 + 32 * ct + acc_col4(g, hh);
                    const uint32_t m = word >> acc_col4(g, hh);
                    float4 v;
                    v.x = (m & 1u) ? acc[rt][ct][4 * g] : 0.f;
                    v.y = (m & 2u) ? acc[rt][ct][4 * g + 1] : 0.f;
                    v.z = (m & 4u) ? acc[rt][ct][4 * g + 2] : 0.f;
                    v.w = (m & 8u) ? acc[rt][ct][4 * g + 3] : 0.f;
                    if (l > 0) store_split4(gP, DF_BF_ACT_PLANE, ((size_t)row * DF_BH + col) * 2, v);
                    *reinterpret_cast<float4*>(a.dz + ((int64_t)l * a.n_pad + p0 + row) * DF_W + col) = v;
                }
            }
        if (l == 0) break;
        __syncthreads();
        load_signs(l - 1);
        zero_acc(acc);
        // segment of W_l in the backward stream: 11 + (7 - l); the walk ends with W_1
        if (l > 1) {
            const WSeg nx = wseg(bf, 11 + (7 - l) + 1, n0 + li, hh);
            stream_gemm_bf<2, NC, DF_W>(acc, g_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, seg, &nx, wcur, wnx1);   // dh_{l-1}
            seg = nx;
        } else {
            stream_gemm_bf<2, NC, DF_W>(acc, g_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, seg, nullptr, wcur, wnx1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Backward walk on TWO fp16 planes (round 6): k_deform_bwd_bf's walk with k_deform_fwd_h's arithmetic -- three
// v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones, two thirds of the operand bytes, 68 KB of LDS (two
// workgroups per CU; the three bf16 planes allowed one).  Weights: the fp16 planes of the backward stream (times 2^10; a
// weight >= 64 sets the pack kernel's flag, this kernel returns and k_deform_bwd_bf behind it does the work).  Gradients
// have no fixed range (1e-12 early in a run, 1e+2 under a large loss scale), so every point's row of the tile carries
// its own power-of-two scale: the largest magnitude of the row goes to [2^13, 2^14) before the split, and the
// accumulator of that point -- a point is a lane of the MFMA's output -- is multiplied by the inverse (times 2^-10 for
// the weights) when it is read.  Powers of two: the scaling itself is exact; entries down to 2^-27 of their row's largest
// keep a normal hi part and 22 bits, smaller ones fade out against a sum (the next layer's gradient, the weight
// gradients) that the large entries of the same row dominate.  Measured against float64: tests/test_deform.py.
// ---------------------------------------------------------------------------------------------
constexpr size_t DF_BWD_H_LDS = 2 * DF_BF_ACT_PLANE + (size_t)(DF_BWD_WAVES + 1) * 64 * 4;   // planes + the rows' maxima per wave + the head tile's

// m = f 2^e with f in [0.5, 1): e (0 for 0, inf, NaN), kept where 2^(14 - e) and 2^(e - 24) are normal floats
__device__ __forceinline__ int grad_exp(float m)
{
    const int e = __builtin_amdgcn_frexp_expf(m);
    return e < -100 ? -100 : e;
}
__device__ __forceinline__ float pow2_f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

__global__ __launch_bounds__(64 * DF_BWD_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_deform_bwd_h(BwdArgs a)
{
    extern __shared__ float4 df_lds[];
    if (__builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.packed + DF_FLAG_OFF)) != 0u) return;
    char* gP = reinterpret_cast<char*>(df_lds);                               // gradient planes [2][64][264] fp16
    float* pm = reinterpret_cast<float*>(gP + 2 * DF_BF_ACT_PLANE);           // [waves][64]: a row's largest magnitude in each wave's columns
    float* hm = pm + DF_BWD_WAVES * 64;                                       // [64]: the same for the head tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p0 = (int64_t)blockIdx.x * 64;
    constexpr int NC = DF_BWD_NC, NT = 64 * DF_BWD_WAVES;
    static_assert(NT == 256, "the head tile's rows are read by 16 consecutive lanes each");
    const int n0 = wave * 32 * NC;
    if (blockIdx.x == 0 && tid == 0) a.xflag[0] = 0u;
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if ((int64_t)blockIdx.x * 64 >= a.plan->n_ext) return;
        a.n = a.plan->n;
    }
    const __amdgpu_buffer_rsrc_t rw = buf_rsrc(a.packed + DF_PACKED_FLOATS + DF_BF_FLOATS, (uint32_t)DF_H_FLOATS * 4u);
    WSegH seg = wseg_h(10, n0 + li, hh);
    uint4 wcur[2][NC], wnx1[2][NC], wnx2[2][NC];
    load_wh<NC, DF_W>(wcur, rw, seg, 0);
    load_wh<NC, DF_W>(wnx1, rw, seg, 1);
    load_wh<NC, DF_W>(wnx2, rw, seg, 2);

    // upstream gradients as the head's output tile: [d_sh (48) | d_xyz (3) | 0], in the first 64 columns; a row is read
    // by 16 consecutive lanes, which agree on its scale among themselves
    {
        float4 hv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (tid >> 4) + 16 * i, col = (tid & 15) * 4;
            const int64_t p = p0 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p < a.n) {
                if (col < 48) {
                    if (a.g_dsh) v = *reinterpret_cast<const float4*>(a.g_dsh + p * 48 + col);
                } else if (col == 48 && a.g_dxyz) {
                    v.x = a.g_dxyz[p * 3]; v.y = a.g_dxyz[p * 3 + 1]; v.z = a.g_dxyz[p * 3 + 2];
                }
            }
            hv[i] = v;
            *reinterpret_cast<float4*>(a.dzh + p * DF_HEAD + col) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = (tid >> 4) + 16 * i, col = (tid & 15) * 4;
            float m = fmaxf(fmaxf(fabsf(hv[i].x), fabsf(hv[i].y)), fmaxf(fabsf(hv[i].z), fabsf(hv[i].w)));
            m = fmaxf(m, __shfl_xor(m, 1));
            m = fmaxf(m, __shfl_xor(m, 2));
            m = fmaxf(m, __shfl_xor(m, 4));
            m = fmaxf(m, __shfl_xor(m, 8));
            const float sc = pow2_f(14 - grad_exp(m));
            float4 v = hv[i];
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            store_split4_h(gP, DF_BF_ACT_PLANE, ((size_t)row * DF_BH + col) * 2, v);
            if ((tid & 15) == 0) hm[row] = m;
        }
    }
    __syncthreads();

    const char* g_lane = gP + (size_t)li * DF_BH * 2 + 16 * hh;
    f32x16 acc[2][NC];
    uint32_t sg[2][NC];
    float inv[2];                      // what turns the accumulator of this lane's two points into the gradient: 1 / (row scale * 2^10)
#pragma unroll
    for (int rt = 0; rt < 2; rt++) inv[rt] = pow2_f(grad_exp(hm[32 * rt + li]) - 24);
    auto load_signs = [&](int l) {
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++)
                sg[rt][ct] = a.signs[((int64_t)l * a.n_pad + p0 + 32 * rt + li) * DF_SIGN_WORDS + NC * wave + ct];
    };
    load_signs(DF_D - 1);
    zero_acc(acc);
    {
        const WSegH nx = wseg_h(11, n0 + li, hh);
        stream_gemm_h<2, NC, DF_W>(acc, g_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_HEAD / 16, rw, seg, &nx, wcur, wnx1, wnx2);   // dh_7
        seg = nx;
    }
    for (int l = DF_D - 1; l >= 0; l--) {
        // dz_l in place of the accumulators, and the largest magnitude of each of the lane's two rows
        float mx[2] = {0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 2; rt++) {
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
                const uint32_t word = sg[rt][ct];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t m = word >> acc_col4(g, hh);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float v = (m & (1u << k)) ? acc[rt][ct][4 * g + k] * inv[rt] : 0.f;
                        acc[rt][ct][4 * g + k] = v;
                        mx[rt] = fmaxf(mx[rt], fabsf(v));
                    }
                }
            }
            mx[rt] = fmaxf(mx[rt], __shfl_xor(mx[rt], 32));
            if (l > 0 && hh == 0) pm[wave * 64 + 32 * rt + li] = mx[rt];
        }
        __syncthreads();      // every wave is past its last read of the previous gradient tile; the rows' maxima are complete
        float sc[2] = {1.f, 1.f};
        if (l > 0) {
#pragma unroll
            for (int rt = 0; rt < 2; rt++) {
                float m = pm[32 * rt + li];
#pragma unroll
                for (int w = 1; w < DF_BWD_WAVES; w++) m = fmaxf(m, pm[w * 64 + 32 * rt + li]);
                const int e = grad_exp(m);
                sc[rt] = pow2_f(14 - e);
                inv[rt] = pow2_f(e - 24);
                if (wave == 0 && hh == 0) a.rowmax[(int64_t)l * a.n_pad + p0 + 32 * rt + li] = m;
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < NC; ct++) {
                const int row = 32 * rt + li;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int col = n0 + 32 * ct + acc_col4(g, hh);
                    const float4 v = make_float4(acc[rt][ct][4 * g], acc[rt][ct][4 * g + 1], acc[rt][ct][4 * g + 2], acc[rt][ct][4 * g + 3]);
                    if (l > 0) {
                        const float4 u = make_float4(v.x * sc[rt], v.y * sc[rt], v.z * sc[rt], v.w * sc[rt]);
                        store_split4_h(gP, DF_BF_ACT_PLANE, ((size_t)row * DF_BH + col) * 2, u);
                    }
                    *reinterpret_cast<float4*>(a.dz + ((int64_t)l * a.n_pad + p0 + row) * DF_W + col) = v;
                }
            }
        if (l == 0) break;
        __syncthreads();
        load_signs(l - 1);
        zero_acc(acc);
        // segment of W_l in the backward stream: 11 + (7 - l); the walk ends with W_1
        if (l > 1) {
            const WSegH nx = wseg_h(11 + (7 - l) + 1, n0 + li, hh);
            stream_gemm_h<2, NC, DF_W>(acc, g_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, rw, seg, &nx, wcur, wnx1, wnx2);   // dh_{l-1}
            seg = nx;
        } else {
            stream_gemm_h<2, NC, DF_W>(acc, g_lane, DF_BH * 2, DF_BF_ACT_PLANE, DF_W / 16, rw, seg, nullptr, wcur, wnx1, wnx2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// weight / bias gradients: dW[out][in] = sum over points of dz[p][out] * x[p][in], split over point
// ranges; operands are read straight from global (lane = consecutive column: coalesced rows)
// ---------------------------------------------------------------------------------------------
constexpr int DW_JOBS = 10;
// partial-sum block of one split (floats)
constexpr int64_t DW_OFF_L0 = 0;                                         // [256][96]
constexpr int64_t DW_OFF_L(int l) { return (int64_t)DF_W * DF_EMB + (int64_t)(l - 1) * DF_F_SZ; }   // l = 1..7: [256][256] (layer 5: hidden part)
constexpr int64_t DW_OFF_L5E = (int64_t)DF_W * DF_EMB + 7 * DF_F_SZ;     // [256][96]
constexpr int64_t DW_OFF_HEAD = DW_OFF_L5E + (int64_t)DF_W * DF_EMB;     // [64][256]
constexpr int64_t DW_OFF_BIAS = DW_OFF_HEAD + (int64_t)DF_HEAD * DF_W;   // [8][256] + [64]
constexpr int64_t DW_PART_FLOATS = DW_OFF_BIAS + DF_D * DF_W + DF_HEAD;

struct DwArgs {
    const DevPlan* plan;    // NULL: the extent is n_pad, the splits are the host's
    int64_t n_pad;          // plane stride of acts / dz / rowmax (the capacity's under a plan)
    int64_t n_ext;          // rows the sums run over (= n_pad without a plan)
    int tiles_per_split;    // 64-point tiles per split
    int splits;
    int first_block;        // the launch covers workgroups first_block ... of the job table (heavy jobs, then light ones)
    const float* emb; const float* acts; const float* dz; const float* dzh;
    float* part;            // [splits][DW_PART_FLOATS]
    const float* packed;    // (the fp16 planes' weight flag)
    const float* rowmax;    // [8][n_pad] from k_deform_bwd_h
    uint32_t* xflag;        // k_deform_dw_h: set when an activation does not fit the planes
    int only_if;            // k_deform_dw_bf's hidden-layer launch behind k_deform_dw_h: run only if that one could not
};

// V consecutive floats as one load / store (V = 2, 3, 4: global_load_dwordx2/x3/x4)
template <int V> struct FVec { float v[V]; };
template <> struct __attribute__((aligned(8))) FVec<2> { float v[2]; };
template <> struct __attribute__((aligned(16))) FVec<4> { float v[4]; };

// One wave's block of dW: 32*VN output rows (n) x 32*VK input columns (k), summed over the points of the range.
// A lane loads VN consecutive columns of dz and VK consecutive columns of x for its point (two vector loads per
// two points instead of VN + VK scalar ones: the loads' address processing, not the multiplies, set the pace
// of the scalar version) and uses element t in tile t, so tile (tn, tk) holds rows n = n_base + VN*i + tn and
// columns k = k_base + VK*j + tk: VK consecutive k per lane and register -> vector stores.
template <int VN, int VK, bool BIAS>
__device__ __forceinline__ void dw_job(const float* A, int lda, int n_base, const float* B, int ldb, int k_base, float* out,
                                       int out_ld, float* bias_out, bool write_bias, int64_t p_begin, int64_t p_end, int li, int hh)
{
    f32x16 acc[VN][VK];
    zero_acc(acc);
    float bsum[VN];
#pragma unroll
    for (int x = 0; x < VN; x++) bsum[x] = 0.f;
    const float* Ap = A + (p_begin + hh) * lda + n_base + VN * li;
    const float* Bp = B + (p_begin + hh) * ldb + k_base + VK * li;
    // operands of four 8-point steps in registers: three steps in flight ahead of the one being multiplied
    FVec<VN> av[4][4];
    FVec<VK> bv[4][4];
    int64_t pf = p_begin;          // first point of the next step to fetch
    auto fetch = [&](int slot) {
        if (pf < p_end) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                av[slot][s] = *reinterpret_cast<const FVec<VN>*>(Ap + (int64_t)2 * s * lda);
                bv[slot][s] = *reinterpret_cast<const FVec<VK>*>(Bp + (int64_t)2 * s * ldb);
            }
            Ap += 8 * lda;
            Bp += 8 * ldb;
        }
        pf += 8;
    };
    auto mul = [&](int slot) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int x = 0; x < VN; x++) {
                if (BIAS) bsum[x] += av[slot][s].v[x];
#pragma unroll
                for (int y = 0; y < VK; y++)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot][s].v[x], bv[slot][s].v[y], acc[x][y], 0, 0, 0);
            }
        }
    };
    // 32 points per loop trip (the ranges are multiples of 64)
    if (p_begin < p_end) {
        fetch(0);
        fetch(1);
        fetch(2);
        for (int64_t p = p_begin; p < p_end; p += 32) {
            fetch(3);
            mul(0);
            fetch(0);
            mul(1);
            fetch(1);
            mul(2);
            fetch(2);
            mul(3);
        }
    }
#pragma unroll
    for (int x = 0; x < VN; x++)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            FVec<VK> o;
#pragma unroll
            for (int y = 0; y < VK; y++) o.v[y] = acc[x][y][q];
            *reinterpret_cast<FVec<VK>*>(out + (int64_t)(n_base + VN * acc_row(q, hh) + x) * out_ld + k_base + VK * li) = o;
        }
    if (BIAS && write_bias) {
#pragma unroll
        for (int x = 0; x < VN; x++) {
            const float tot = bsum[x] + __shfl_xor(bsum[x], 32);
            if (hh == 0) bias_out[n_base + VN * li + x] = tot;
        }
    }
}

// The same block of dW from six bf16 MFMAs per product: a lane loads, per 16-point step, the eight values of its
// output row (column of dz) and of its input column (column of x) at its eight points -- v_mfma_f32_32x32x16_bf16
// wants eight consecutive k per lane, k = point here -- splits them into hi / mid / lo and multiplies; the
// operands stay fp32 in memory (4 B per element from HBM instead of 6), the split costs VALU time between the
// multiplies of consecutive steps.
template <int TN, int TK, bool BIAS>
__device__ __forceinline__ void dw_job_bf(const float* A, int lda, int n_base, const float* B, int ldb, int k_base, float* out,
                                          int out_ld, float* bias_out, bool write_bias, int64_t p_begin, int64_t p_end, int li, int hh)
{
    // The light jobs (first layer, encoding rows of layer 5, heads): every wave loads and splits the blocks of both
    // operands it multiplies.  Same pipeline as dw_job_bf_shared, in registers: two steps of loads in flight (rawA /
    // rawB), the planes of step k + 1 are split (two values per v_cvt_pk_bf16_f32) between the multiplies of step k,
    // and every use of a raw value stays in front of the load that refills its register -- before, the copies at the
    // end of the loop body waited for the loads of the step that had just been requested, every step.
    constexpr int NB = TN + TK;                      // 32-column blocks of both operands
    constexpr int NPAIR = NB * 4, NMUL = 6 * TN * TK;
    f32x16 acc[TN][TK];
    zero_acc(acc);
    float bsum[TN];
#pragma unroll
    for (int x = 0; x < TN; x++) bsum[x] = 0.f;
    const int64_t nsteps = (p_end - p_begin) / 16;   // a multiple of 4 (the ranges are multiples of 64 points)
    const float* Ap = A + (p_begin + 8 * hh) * lda + n_base + li;
    const float* Bp = B + (p_begin + 8 * hh) * ldb + k_base + li;
    const float* Ap_last = Ap + (nsteps > 0 ? nsteps - 1 : 0) * 16 * lda;
    const float* Bp_last = Bp + (nsteps > 0 ? nsteps - 1 : 0) * 16 * ldb;
    float rawA[NB][8], rawB[NB][8];
    uint32_t plA[NB][3][4], plB[NB][3][4];           // the planes of an even / odd step: 8 bf16 per lane, block and plane
    auto load1 = [&](int blk, int j) -> float {
        return blk < TN ? Ap[(int64_t)j * lda + 32 * blk] : Bp[(int64_t)j * ldb + 32 * (blk - TN)];
    };
    auto advance = [&]() {
        Ap += 16 * lda;
        Bp += 16 * ldb;
        Ap = Ap > Ap_last ? Ap_last : Ap;
        Bp = Bp > Bp_last ? Bp_last : Bp;
    };
    auto fetch = [&](float (&raw)[NB][8]) {
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int blk = 0; blk < NB; blk++) raw[blk][j] = load1(blk, j);
        advance();
    };
    // splits the pair (j, j + 1) of block blk into the planes `pf`; refill: loads the pair of the step Ap / Bp point at
    auto split_pair = [&](float (&raw)[NB][8], uint32_t (&pf)[NB][3][4], int blk, int j, bool live, bool refill) {
        const float v0 = raw[blk][j], v1 = raw[blk][j + 1];
        if (BIAS && blk < TN && write_bias) {          // (wave-uniform: the waves of the other column half leave the sums out)
            bsum[blk < TN ? blk : 0] += live ? v0 : 0.f;
            bsum[blk < TN ? blk : 0] += live ? v1 : 0.f;
        }
        const uint32_t ph = cvt_pk_bf16(v0, v1);
        const float r0 = v0 - __uint_as_float(ph << 16), r1 = v1 - __uint_as_float(ph & 0xffff0000u);
        const uint32_t pm = cvt_pk_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
        pf[blk][0][j >> 1] = ph;
        pf[blk][1][j >> 1] = pm;
        pf[blk][2][j >> 1] = cvt_pk_bf16(s0, s1);
        if (refill) {
            asm volatile("" : "+v"(pf[blk][0][j >> 1]), "+v"(pf[blk][1][j >> 1]), "+v"(pf[blk][2][j >> 1]), "+v"(bsum[blk < TN ? blk : 0])
                         : : "memory");
            raw[blk][j] = load1(blk, j);
            raw[blk][j + 1] = load1(blk, j + 1);
        }
    };
    auto plane = [&](const uint32_t (&pl)[NB][3][4], int blk, int q) -> bf16x8 {
        const uint4 v = make_uint4(pl[blk][q][0], pl[blk][q][1], pl[blk][q][2], pl[blk][q][3]);
        return as_bf(v);
    };
    if (nsteps > 0) {
        fetch(rawA);                                   // step 0
#pragma unroll
        for (int c = 0; c < NPAIR; c++) split_pair(rawA, plA, c >> 2, (c & 3) * 2, true, false);
        fetch(rawA);                                   // step 1
        fetch(rawB);                                   // step 2; Ap / Bp now at step 3
    }
    // one step: multiplies with the planes `pu` of step k; `raw` holds step k + 1, is split into `pf` and refilled
    // with step k + 3
    auto step = [&](int64_t k, float (&raw)[NB][8], const uint32_t (&pu)[NB][3][4], uint32_t (&pf)[NB][3][4]) {
        const bool live = k + 1 < nsteps;              // (behind the end: planes that are never used, no bias share)
#pragma unroll
        for (int c = 0; c < NPAIR; c++) {
#pragma unroll
            for (int m = c * NMUL / NPAIR; m < (c + 1) * NMUL / NPAIR; m++) {
                const int term = m / (TN * TK), x = (m / TK) % TN, y = m % TK;
                const int pw = term == 0 ? 0 : term == 1 ? 1 : term == 2 ? 0 : term == 3 ? 2 : term == 4 ? 0 : 1;
                const int pv = term == 0 ? 0 : term == 1 ? 0 : term == 2 ? 1 : term == 3 ? 0 : term == 4 ? 2 : 1;
                acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(plane(pu, x, pw), plane(pu, TN + y, pv), acc[x][y], 0, 0, 0);
            }
            split_pair(raw, pf, c >> 2, (c & 3) * 2, live, true);
            __builtin_amdgcn_sched_barrier(0);
        }
        advance();
    };
    for (int64_t k = 0; k < nsteps; k += 2) {
        step(k, rawA, plA, plB);
        step(k + 1, rawB, plB, plA);
    }
#pragma unroll
    for (int x = 0; x < TN; x++)
#pragma unroll
        for (int y = 0; y < TK; y++)
#pragma unroll
            for (int q = 0; q < 16; q++)
                out[(int64_t)(n_base + 32 * x + acc_row(q, hh)) * out_ld + k_base + 32 * y + li] = acc[x][y][q];
    if (BIAS && write_bias) {
#pragma unroll
        for (int x = 0; x < TN; x++) {
            const float tot = bsum[x] + __shfl_xor(bsum[x], 32);
            if (hh == 0) bias_out[n_base + 32 * x + li] = tot;
        }
    }
}

// The 256 x 256 block of a hidden layer's dW by one workgroup, every operand block split ONCE: in dw_job_bf each of the
// four waves loads and splits both of its 128-column operand blocks, so every block is split by two waves and the
// splitting takes as long as the multiplies.  Here wave w loads and splits one block (w = 0, 1: columns 0..127 / 128..255
// of dz; w = 2, 3: of x), leaves the three bf16 planes in LDS in the lanes' MFMA operand layout, and after a barrier
// every wave reads the planes of its two blocks (16-byte LDS reads, the same lane slot as the writer's).  Two LDS
// buffers: the planes of step k + 1 are written while the multiplies of step k run; one barrier per 16-point step.  The
// same products in the same order as dw_job_bf: bit-identical dW.
constexpr int DW_SH_BLOCK_BYTES = 4 * 3 * 64 * 16;                 // [x 4][plane 3][lane 64] x 16 B
constexpr int DW_SH_LDS = 2 * 4 * DW_SH_BLOCK_BYTES;               // two buffers of four operand blocks: 98304 B
__device__ __forceinline__ void dw_job_bf_shared(const float* A, const float* B, float* out, float* bias_out, int64_t p_begin,
                                                 int64_t p_end, int wave, int lane, char* lds)
{
    const int li = lane & 31, hh = lane >> 5;
    f32x16 acc[4][4];
    zero_acc(acc);
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    // the block this wave loads and splits
    const float* S = (wave < 2 ? A : B) + (p_begin + 8 * hh) * DF_W + 128 * (wave & 1) + li;
    const bool sum_bias = wave < 2;
    const int64_t nsteps = (p_end - p_begin) / 16;       // a multiple of 4 (the ranges are multiples of 64 points)
    // Measured at 300 k points, hidden layers only: operand loads alone 0.84 ms (4.3 GB of dz and x, 5.1 TB/s), loads +
    // split + LDS 0.89 ms, everything 1.28 ms; the 96 multiplies of a step are 3072 cycles of the SIMD's matrix pipe.
    // Two steps of loads are kept in flight per wave (rawA / rawB, 16 KB per wave).  The loop body is one basic block
    // -- the steps behind the end split and fetch the last step again instead of branching.
    const float* S_last = S + (nsteps > 0 ? nsteps - 1 : 0) * 16 * DF_W;
    float rawA[4][8], rawB[4][8];
    auto fetch = [&](float (&raw)[4][8]) {
        S = S > S_last ? S_last : S;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int x = 0; x < 4; x++) raw[x][j] = S[(int64_t)j * DF_W + 32 * x];
        S += 16 * DF_W;
    };
    auto split_to = [&](char* buf, const float (&raw)[4][8], bool live) {
        uint4* dst = reinterpret_cast<uint4*>(buf + wave * DW_SH_BLOCK_BYTES) + lane;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            float add = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) add += raw[x][j];
            bsum[x] += (sum_bias && live) ? add : 0.f;
            bf16x8 pl[3];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                __bf16 h, m, l;
                split3(raw[x][j], h, m, l);
                pl[0][j] = h; pl[1][j] = m; pl[2][j] = l;
            }
#pragma unroll
            for (int q = 0; q < 3; q++) {
                uint4 v;
                __builtin_memcpy(&v, &pl[q], 16);
                dst[(x * 3 + q) * 64] = v;
            }
        }
    };
    if (nsteps > 0) {
        fetch(rawA);                        // step 0
        split_to(lds, rawA, true);
        fetch(rawA);                        // step 1
        fetch(rawB);                        // step 2
    }
    __syncthreads();
    const int ablk = wave & 1, bblk = 2 + (wave >> 1);
    // One step: the planes of step k are in buffer k & 1; `raw` holds step k + 1 and is refilled with step k + 3.
    // The order is written out and fenced (sched_barrier) chunk by chunk -- left to itself the scheduler puts the 96
    // multiplies of every other step in one run with the split behind them, and the step takes the sum of the two
    // instead of the longer one.  A chunk = three multiplies, the split of one raw value per lane, the load that
    // refills it, and its share of the LDS traffic: the planes of the later terms are read while the earlier terms
    // multiply, the planes of a finished 32-column block are written as soon as its eighth value is split.
    auto step = [&](int64_t k, float (&raw)[4][8]) {
        char* cur = lds + (k & 1) * 4 * DW_SH_BLOCK_BYTES;
        char* nxt = lds + ((k + 1) & 1) * 4 * DW_SH_BLOCK_BYTES;
        const uint4* sa = reinterpret_cast<const uint4*>(cur + ablk * DW_SH_BLOCK_BYTES) + lane;
        const uint4* sb = reinterpret_cast<const uint4*>(cur + bblk * DW_SH_BLOCK_BYTES) + lane;
        uint4* dst = reinterpret_cast<uint4*>(nxt + wave * DW_SH_BLOCK_BYTES) + lane;
        const bool live = k + 1 < nsteps;        // (behind the end: planes that are never read, no bias share)
        S = S > S_last ? S_last : S;
        bf16x8 pa[4][3], pb[4][3];
        auto read_a = [&](int q) {
#pragma unroll
            for (int x = 0; x < 4; x++) { const uint4 v = sa[(x * 3 + q) * 64]; __builtin_memcpy(&pa[x][q], &v, 16); }
        };
        auto read_b = [&](int q) {
#pragma unroll
            for (int x = 0; x < 4; x++) { const uint4 v = sb[(x * 3 + q) * 64]; __builtin_memcpy(&pb[x][q], &v, 16); }
        };
        read_a(0);
        read_b(0);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t pl[3][4];                       // the three planes of one 32-column block: 8 bf16 per lane each
        float add = 0.f;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int x = c >> 2, j = (c & 3) * 2;
            if (c == 0) read_a(1);               // term 1 (from multiply 16)
            if (c == 2) read_b(1);               // term 2 (from 32)
            if (c == 5) read_a(2);               // term 3 (from 48)
            if (c == 8) read_b(2);               // term 4 (from 64)
#pragma unroll
            for (int m = 6 * c; m < 6 * c + 6; m++) {
                const int term = m >> 4, mx = (m >> 2) & 3, my = m & 3;
                const int pw = term == 0 ? 0 : term == 1 ? 1 : term == 2 ? 0 : term == 3 ? 2 : term == 4 ? 0 : 1;
                const int pv = term == 0 ? 0 : term == 1 ? 0 : term == 2 ? 1 : term == 3 ? 0 : term == 4 ? 2 : 1;
                acc[mx][my] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[mx][pw], pb[my][pv], acc[mx][my], 0, 0, 0);
            }
            {
                // split3 of two values at once: v_cvt_pk_bf16_f32 rounds both and leaves them packed as the planes
                // want them (13 VALU instructions per pair, bias sum included)
                const float v0 = raw[x][j], v1 = raw[x][j + 1];
                add += v0;
                add += v1;
                const uint32_t ph = cvt_pk_bf16(v0, v1);
                const float r0 = v0 - __uint_as_float(ph << 16), r1 = v1 - __uint_as_float(ph & 0xffff0000u);
                const uint32_t pm = cvt_pk_bf16(r0, r1);
                const float s0 = r0 - __uint_as_float(pm << 16), s1 = r1 - __uint_as_float(pm & 0xffff0000u);
                pl[0][j >> 1] = ph;
                pl[1][j >> 1] = pm;
                pl[2][j >> 1] = cvt_pk_bf16(s0, s1);
                // every use of the old values stays in front of the loads that refill their registers (instruction
                // selection is free to put them behind otherwise; old and new value then need two registers, and the
                // copies at the end of the loop body wait for every load in flight)
                asm volatile("" : "+v"(add), "+v"(pl[0][j >> 1]), "+v"(pl[1][j >> 1]), "+v"(pl[2][j >> 1]) : : "memory");
                raw[x][j] = S[(int64_t)j * DF_W + 32 * x];
                raw[x][j + 1] = S[(int64_t)(j + 1) * DF_W + 32 * x];
            }
            if (j == 6) {
                bsum[x] += (sum_bias && live) ? add : 0.f;
                add = 0.f;
#pragma unroll
                for (int q = 0; q < 3; q++) dst[(x * 3 + q) * 64] = make_uint4(pl[q][0], pl[q][1], pl[q][2], pl[q][3]);
            }
            // one multiply, then at most three other instructions (a wave alone on its SIMD hides about 24 cycles
            // of issue behind each 32-cycle multiply)
#pragma unroll
            for (int i = 0; i < 6; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // VALU
                if (i == 1 || i == 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
                if (i == 2 || i == 4) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   // DS write
                if (i == 5) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);             // VMEM read
                if (i == 0 || i == 5) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        S += 16 * DF_W;
        __syncthreads();      // everyone has read `cur` and written `nxt`
    };
    for (int64_t k = 0; k < nsteps; k += 2) {
        step(k, rawA);
        step(k + 1, rawB);
    }
    const int n_base = 128 * (wave & 1), k_base = 128 * (wave >> 1);
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int q = 0; q < 16; q++)
                out[(int64_t)(n_base + 32 * x + acc_row(q, hh)) * DF_W + k_base + 32 * y + li] = acc[x][y][q];
    if (sum_bias) {
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const float tot = bsum[x] + __shfl_xor(bsum[x], 32);
            if (hh == 0) bias_out[n_base + 32 * x + li] = tot;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The hidden layers' dW on TWO fp16 planes (round 6): dw_job_bf_shared's pipeline with three multiplies per product.
// Scales: an activation goes into the planes times 2^4 as in the forward walk (one beyond 4094 raises `xflag` and the
// bf16 job launched behind this kernel redoes the work); a gradient has no fixed range, and the sum runs over the
// points, so the scale must be one per accumulation: the workgroup takes the largest |dz| of ITS points from the row
// maxima k_deform_bwd_h left (a power of two that puts it in [2^13, 2^14)) and divides its partial sums by it.  Two
// planes: 64 KB of LDS instead of 96 (still one workgroup per CU: its waves hold 256 accumulator registers each).
// ---------------------------------------------------------------------------------------------
constexpr int DW_H_BLOCK_BYTES = 4 * 2 * 64 * 16;                  // [x 4][plane 2][lane 64] x 16 B
constexpr int DW_H_LDS = 2 * 4 * DW_H_BLOCK_BYTES + 64;            // two buffers of four operand blocks + the scale exchange: 65600 B
template <int DEPTH>      // 16-point steps of operand loads in flight per wave: 2 or 3
__device__ __forceinline__ void dw_job_h_shared(const float* A, const float* B, const float* rowmax, float* out, float* bias_out,
                                                int64_t p_begin, int64_t p_end, int wave, int lane, char* lds, uint32_t* xflag)
{
    const int li = lane & 31, hh = lane >> 5;
    // the scale of this workgroup's gradients
    float* ex = reinterpret_cast<float*>(lds + 2 * 4 * DW_H_BLOCK_BYTES);
    {
        float m = 0.f;
        for (int64_t p = p_begin + wave * 64 + lane; p < p_end; p += 256) m = fmaxf(m, rowmax[p]);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
        if (lane == 0) ex[wave] = m;
    }
    __syncthreads();
    const int e_dz = grad_exp(fmaxf(fmaxf(ex[0], ex[1]), fmaxf(ex[2], ex[3])));
    const float scale = wave < 2 ? pow2_f(14 - e_dz) : DF_H_ASCALE;
    f32x16 acc[4][4];
    zero_acc(acc);
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    float top = 0.f;
    const float* S = (wave < 2 ? A : B) + (p_begin + 8 * hh) * DF_W + 128 * (wave & 1) + li;
    const bool sum_bias = wave < 2;
    const int64_t nsteps = (p_end - p_begin) / 16;       // a multiple of 4 (the ranges are multiples of 64 points)
    const float* S_last = S + (nsteps > 0 ? nsteps - 1 : 0) * 16 * DF_W;
    float rawA[4][8], rawB[4][8], rawC[4][8];
    auto fetch = [&](float (&raw)[4][8]) {
        S = S > S_last ? S_last : S;
#pragma unroll
        for (int j = 0; j < 8; j++)
#pragma unroll
            for (int x = 0; x < 4; x++) raw[x][j] = S[(int64_t)j * DF_W + 32 * x];
        S += 16 * DF_W;
    };
    // a pair of values -> their packed hi and lo halves
    auto split2 = [&](float v0, float v1, uint32_t& hi, uint32_t& lo) {
        const float s0 = v0 * scale, s1 = v1 * scale;
        top = fmaxf(top, fmaxf(fabsf(s0), fabsf(s1)));
        hi = cvt_pk_f16(s0, s1);
        lo = cvt_pk_f16(sub_f16_lo(s0, hi), sub_f16_hi(s1, hi));
    };
    if (nsteps > 0) {
        fetch(rawA);                        // step 0
        uint4* dst = reinterpret_cast<uint4*>(lds + wave * DW_H_BLOCK_BYTES) + lane;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            uint32_t ph[4], pl[4];
            float add = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                add += rawA[x][j] + rawA[x][j + 1];
                split2(rawA[x][j], rawA[x][j + 1], ph[j >> 1], pl[j >> 1]);
            }
            bsum[x] += sum_bias ? add : 0.f;
            dst[(x * 2 + 0) * 64] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
            dst[(x * 2 + 1) * 64] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
        }
        fetch(rawA);                        // step 1
        fetch(rawB);                        // step 2
        if (DEPTH == 3) fetch(rawC);        // step 3
    }
    __syncthreads();
    const int ablk = wave & 1, bblk = 2 + (wave >> 1);
    // One step: the planes of step k are in buffer k & 1; `raw` holds step k + 1 and is refilled with step k + 1 + DEPTH.  A chunk =
    // three multiplies, the split of one pair of raw values, the loads that refill them and a share of the LDS traffic.
    auto step = [&](int64_t k, float (&raw)[4][8]) {
        char* cur = lds + (k & 1) * 4 * DW_H_BLOCK_BYTES;
        char* nxt = lds + ((k + 1) & 1) * 4 * DW_H_BLOCK_BYTES;
        const uint4* sa = reinterpret_cast<const uint4*>(cur + ablk * DW_H_BLOCK_BYTES) + lane;
        const uint4* sb = reinterpret_cast<const uint4*>(cur + bblk * DW_H_BLOCK_BYTES) + lane;
        uint4* dst = reinterpret_cast<uint4*>(nxt + wave * DW_H_BLOCK_BYTES) + lane;
        const bool live = k + 1 < nsteps;        // (behind the end: planes that are never read, no bias share)
        S = S > S_last ? S_last : S;
        f16x8 pa[4][2], pb[4][2];
        auto read_a = [&](int q) {
#pragma unroll
            for (int x = 0; x < 4; x++) { const uint4 v = sa[(x * 2 + q) * 64]; __builtin_memcpy(&pa[x][q], &v, 16); }
        };
        auto read_b = [&](int q) {
#pragma unroll
            for (int x = 0; x < 4; x++) { const uint4 v = sb[(x * 2 + q) * 64]; __builtin_memcpy(&pb[x][q], &v, 16); }
        };
        read_a(0);
        read_b(0);
        __builtin_amdgcn_sched_barrier(0);
        uint32_t ph[4], pl[4];                   // the two planes of one 32-column block: 8 fp16 per lane each
        float add = 0.f;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int x = c >> 2, j = (c & 3) * 2;
            if (c == 0) read_b(1);               // term 1: hi x lo (from multiply 16)
            if (c == 4) read_a(1);               // term 2: lo x hi (from 32)
#pragma unroll
            for (int m = 3 * c; m < 3 * c + 3; m++) {
                const int term = m >> 4, mx = (m >> 2) & 3, my = m & 3;
                acc[mx][my] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pa[mx][term == 2 ? 1 : 0], pb[my][term == 1 ? 1 : 0], acc[mx][my], 0, 0, 0);
            }
            {
                const float v0 = raw[x][j], v1 = raw[x][j + 1];
                add += v0;
                add += v1;
                split2(v0, v1, ph[j >> 1], pl[j >> 1]);
                // every use of the old values stays in front of the loads that refill their registers
                asm volatile("" : "+v"(add), "+v"(ph[j >> 1]), "+v"(pl[j >> 1]), "+v"(top) : : "memory");
                raw[x][j] = S[(int64_t)j * DF_W + 32 * x];
                raw[x][j + 1] = S[(int64_t)(j + 1) * DF_W + 32 * x];
            }
            if (j == 6) {
                bsum[x] += (sum_bias && live) ? add : 0.f;
                add = 0.f;
                dst[(x * 2 + 0) * 64] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                dst[(x * 2 + 1) * 64] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
                if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
                if (i == 1) __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);   // DS write
                if (i == 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        S += 16 * DF_W;
        __syncthreads();      // everyone has read `cur` and written `nxt`
    };
    // Two or three steps of loads in flight per wave.  Measured (MI355X, hidden layers only): at 300 k points the kernel
    // moves its 4.3 GB at 5.3 TB/s with two and a third changes nothing (809 / 830 us); at 100 k points -- 511 workgroups in
    // two rounds, a step = one loaded memory round trip over the steps in flight -- three take 335 us where two take 364.
    // With three the buffers rotate and the last one or two steps follow the loop (nsteps is a multiple of 4, not of 3).
    if (DEPTH == 3) {
        int64_t k = 0;
        for (; k + 3 <= nsteps; k += 3) {
            step(k, rawA);
            step(k + 1, rawB);
            step(k + 2, rawC);
        }
        if (k < nsteps) step(k, rawA);
        if (k + 1 < nsteps) step(k + 1, rawB);
    } else {
        for (int64_t k = 0; k < nsteps; k += 2) {
            step(k, rawA);
            step(k + 1, rawB);
        }
    }
    // (an activation beyond the planes: NaN counts -- the comparison is false for it)
    if (!sum_bias && !(top <= DF_H_MAX)) __hip_atomic_store(xflag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float inv = pow2_f(e_dz - 18);          // 1 / (2^(14 - e) 2^4)
    const int n_base = 128 * (wave & 1), k_base = 128 * (wave >> 1);
#pragma unroll
    for (int x = 0; x < 4; x++)
#pragma unroll
        for (int y = 0; y < 4; y++)
#pragma unroll
            for (int q = 0; q < 16; q++)
                out[(int64_t)(n_base + 32 * x + acc_row(q, hh)) * DF_W + k_base + 32 * y + li] = acc[x][y][q] * inv;
    if (sum_bias) {
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const float tot = bsum[x] + __shfl_xor(bsum[x], 32);
            if (hh == 0) bias_out[n_base + 32 * x + li] = tot;
        }
    }
}

template <int DEPTH>
__global__ __launch_bounds__(256) void k_deform_dw_h(DwArgs a)
{
    extern __shared__ float4 df_lds[];
    if (__builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.packed + DF_FLAG_OFF)) != 0u) return;
    const int wg = (int)blockIdx.x;
    const int job = wg % 7, split = wg / 7;
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return
        if (split >= a.plan->splits) return;
        a.tiles_per_split = a.plan->tiles_per_split; a.n_ext = a.plan->n_ext;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t p_begin = (int64_t)split * a.tiles_per_split * DF_DW_TILE;
    int64_t p_end = p_begin + (int64_t)a.tiles_per_split * DF_DW_TILE;
    if (p_end > a.n_ext) p_end = a.n_ext;
    float* part = a.part + (int64_t)split * DW_PART_FLOATS;
    const int64_t plane = a.n_pad * DF_W;
    const int l = 7 - job;
    dw_job_h_shared<DEPTH>(a.dz + l * plane, a.acts + (l - 1) * plane, a.rowmax + (int64_t)l * a.n_pad, part + DW_OFF_L(l), part + DW_OFF_BIAS + l * DF_W,
                    p_begin, p_end, wave, lane, reinterpret_cast<char*>(df_lds), a.xflag);
}

__global__ __launch_bounds__(256) void k_deform_dw_bf(DwArgs a)
{
    if (a.only_if && __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(a.packed + DF_FLAG_OFF)) == 0u &&
        __hip_atomic_load(a.xflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
        return;
    int job, split;
    if (a.plan) {      // counts the device keeps (DevPlan): the launch is the capacity's, surplus workgroups return below
        a.splits = a.plan->splits; a.tiles_per_split = a.plan->tiles_per_split; a.n_ext = a.plan->n_ext;
        if (a.first_block) a.first_block = 7 * a.splits;      // (the light jobs' launch: behind the heavy jobs of THESE splits)
    }
    const int wg = (int)blockIdx.x + a.first_block;
    if (wg < 7 * a.splits) {
        job = wg % 7;
        split = wg / 7;
    } else {
        const int r = wg - 7 * a.splits;
        job = 7 + r % 3;
        split = r / 3;
    }
    if (split >= a.splits) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p_begin = (int64_t)split * a.tiles_per_split * DF_DW_TILE;
    int64_t p_end = p_begin + (int64_t)a.tiles_per_split * DF_DW_TILE;
    if (p_end > a.n_ext) p_end = a.n_ext;
    float* part = a.part + (int64_t)split * DW_PART_FLOATS;
    const int64_t plane = a.n_pad * DF_W;
    if (job < 7) {
        const int l = 7 - job;
        extern __shared__ float4 df_lds[];
        dw_job_bf_shared(a.dz + l * plane, a.acts + (l - 1) * plane, part + DW_OFF_L(l), part + DW_OFF_BIAS + l * DF_W, p_begin,
                         p_end, wave, lane, reinterpret_cast<char*>(df_lds));
    } else if (job == 7) {
        dw_job_bf<2, 3, true>(a.dz, DF_W, 64 * wave, a.emb, DF_EMB, 0, part + DW_OFF_L0, DF_EMB, part + DW_OFF_BIAS, true, p_begin,
                              p_end, li, hh);
    } else if (job == 8) {
        dw_job_bf<2, 3, false>(a.dz + 5 * plane, DF_W, 64 * wave, a.emb, DF_EMB, 0, part + DW_OFF_L5E, DF_EMB, nullptr, false,
                               p_begin, p_end, li, hh);
    } else {
        dw_job_bf<2, 2, true>(a.dzh, DF_HEAD, 0, a.acts + 7 * plane, DF_W, 64 * wave, part + DW_OFF_HEAD, DF_W,
                              part + DW_OFF_BIAS + DF_D * DF_W, wave == 0, p_begin, p_end, li, hh);
    }
}

__global__ __launch_bounds__(256) void k_deform_dw(DwArgs a)
{
    // heavy jobs first: the 7 hidden-layer GEMMs of every split (7 x splits workgroups, one resident per CU at
    // a time), then the three light jobs, which fill the CUs as they run out of heavy ones
    int job, split;
    if (a.plan) { a.splits = a.plan->splits; a.tiles_per_split = a.plan->tiles_per_split; a.n_ext = a.plan->n_ext; }
    if ((int)blockIdx.x < 7 * a.splits) {
        job = blockIdx.x % 7;
        split = blockIdx.x / 7;
    } else {
        const int r = blockIdx.x - 7 * a.splits;
        job = 7 + r % 3;
        split = r / 3;
    }
    if (split >= a.splits) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, hh = lane >> 5;
    const int64_t p_begin = (int64_t)split * a.tiles_per_split * DF_DW_TILE;
    int64_t p_end = p_begin + (int64_t)a.tiles_per_split * DF_DW_TILE;
    if (p_end > a.n_ext) p_end = a.n_ext;
    float* part = a.part + (int64_t)split * DW_PART_FLOATS;
    const int64_t plane = a.n_pad * DF_W;
    if (job < 7) {
        // hidden-input layers 7..1 (heaviest first): x = act_{l-1}; waves take the four 128 x 128 quadrants
        const int l = 7 - job;
        dw_job<4, 4, true>(a.dz + l * plane, DF_W, 128 * (wave & 1), a.acts + (l - 1) * plane, DF_W, 128 * (wave >> 1),
                           part + DW_OFF_L(l), DF_W, part + DW_OFF_BIAS + l * DF_W, (wave >> 1) == 0, p_begin, p_end, li, hh);
    } else if (job == 7) {
        // layer 0 and the encoding rows of layer 5: x = encoding (96 stored columns); waves take 64 output rows each
        dw_job<2, 3, true>(a.dz, DF_W, 64 * wave, a.emb, DF_EMB, 0, part + DW_OFF_L0, DF_EMB, part + DW_OFF_BIAS, true, p_begin,
                           p_end, li, hh);
    } else if (job == 8) {
        dw_job<2, 3, false>(a.dz + 5 * plane, DF_W, 64 * wave, a.emb, DF_EMB, 0, part + DW_OFF_L5E, DF_EMB, nullptr, false,
                            p_begin, p_end, li, hh);
    } else {
        // heads: 64 output rows, waves take 64 input columns each
        dw_job<2, 2, true>(a.dzh, DF_HEAD, 0, a.acts + 7 * plane, DF_W, 64 * wave, part + DW_OFF_HEAD, DF_W,
                           part + DW_OFF_BIAS + DF_D * DF_W, wave == 0, p_begin, p_end, li, hh);
    }
}

// sum of the splits, written in torch's layouts
struct ReduceSeg {
    float* dst;
    int rows, cols, dst_ld, dst_col0;
    int64_t src_off;
    int src_ld, row_mul, row_add;     // source row = row * row_mul + row_add
};
constexpr int DF_MAX_SEGS = 32;
struct ReduceArgs {
    const DevPlan* plan;    // NULL: splits is the host's
    const float* part;
    int splits, nseg;
    ReduceSeg seg[DF_MAX_SEGS];
};

__global__ __launch_bounds__(256) void k_deform_reduce(ReduceArgs a)
{
    const ReduceSeg& sg = a.seg[blockIdx.y];
    const int splits = a.plan ? a.plan->splits : a.splits;      // (DevPlan: the splits that ran)
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)sg.rows * sg.cols) return;
    const int row = (int)(e / sg.cols), col = (int)(e - (int64_t)row * sg.cols);
    const float* src = a.part + sg.src_off + (int64_t)(row * sg.row_mul + sg.row_add) * sg.src_ld + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
        s0 += src[(int64_t)s * DW_PART_FLOATS];
        s1 += src[(int64_t)(s + 1) * DW_PART_FLOATS];
        s2 += src[(int64_t)(s + 2) * DW_PART_FLOATS];
        s3 += src[(int64_t)(s + 3) * DW_PART_FLOATS];
    }
    for (; s < splits; s++) s0 += src[(int64_t)s * DW_PART_FLOATS];
    sg.dst[(int64_t)row * sg.dst_ld + sg.dst_col0 + col] = (s0 + s1) + (s2 + s3);
}

__host__ __device__ int64_t pad_points(int64_t n) { return (n + DF_PAD - 1) / DF_PAD * DF_PAD; }

__host__ __device__ int dw_splits(int64_t n_pad, int* tiles_per_split)
{
    const int64_t tiles = n_pad / DF_DW_TILE;
    // 7 heavy jobs per split: 73 splits = 511 workgroups = 2 per CU (measured at 300 k points against 146 / 109 splits:
    // hidden-layer jobs 1.18-1.22 vs 1.21-1.26 ms, partial sums to reduce 35 vs 68 us, light jobs 0.27 vs 0.29 ms)
    int64_t splits = tiles / 4;
    if (splits < 1) splits = 1;
    if (splits > 73) splits = 73;
    const int64_t tps = (tiles + splits - 1) / splits;
    *tiles_per_split = (int)tps;
    return (int)((tiles + tps - 1) / tps);
}

// GFT_DEFORM_BF16X3=0: the walks multiply with v_mfma_f32_32x32x2_f32 instead of six bf16 MFMAs per product
bool bf16_planes()
{
    static const bool on = [] { const char* e = getenv("GFT_DEFORM_BF16X3"); return e ? atoi(e) != 0 : true; }();
    return on;
}
// GFT_DEFORM_FP16X2=0: the forward walk on three bf16 planes (six multiplies per product) instead of two fp16 planes (three)
bool fp16_forward()
{
    static const bool on = [] { const char* e = getenv("GFT_DEFORM_FP16X2"); return e ? atoi(e) != 0 : true; }();
    return on;
}

// GFT_DEFORM_BWD_FP16=0: the backward walk on three bf16 planes (six multiplies per product) instead of two fp16 planes
bool fp16_backward()
{
    static const bool on = [] { const char* e = getenv("GFT_DEFORM_BWD_FP16"); return e ? atoi(e) != 0 : true; }();
    return on;
}

// the walks' dynamic-LDS opt-in, per device (gft_lds_opt_in)
hipError_t set_attrs()
{
    static std::atomic<uint64_t> done[9];
    struct { const void* fn; size_t bytes; } k[9] = {
        {reinterpret_cast<const void*>(&k_deform_bwd_h), DF_BWD_H_LDS},
        {reinterpret_cast<const void*>(&k_deform_fwd_h<true, DF_FWD_H_SAVE_WAVES>), DF_FWD_H_SAVE_LDS},
        {reinterpret_cast<const void*>(&k_deform_fwd_h<false, DF_FWD_WAVES>), DF_FWD_H_LDS},
        {reinterpret_cast<const void*>(&k_deform_bwd_bf), DF_BWD_BF_LDS},
        {reinterpret_cast<const void*>(&k_deform_fwd_bf<true>), DF_FWD_BF_LDS},
        {reinterpret_cast<const void*>(&k_deform_fwd_bf<false>), DF_FWD_BF_LDS},
        {reinterpret_cast<const void*>(&k_deform_fwd<true>), DF_FWD_LDS},
        {reinterpret_cast<const void*>(&k_deform_fwd<false>), DF_FWD_LDS},
        {reinterpret_cast<const void*>(&k_deform_bwd), DF_BWD_LDS}};
    for (int i = 0; i < 9; i++) {
        const hipError_t e = gft_lds_opt_in(k[i].fn, k[i].bytes, done[i]);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- backward over the rows with a non-zero upstream gradient only (gft_deform_compact) ---------------------------
// idx[rank[row]] = row for the selected rows
__global__ __launch_bounds__(256) void k_deform_compact_index(int64_t n, const uint8_t* __restrict__ mask, const int32_t* __restrict__ rank,
                                                              int32_t* __restrict__ idx)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row < n && mask[row]) idx[rank[row]] = (int32_t)row;
}

// dst[plane][r][0..width) = src[plane][idx[r]][0..width) for r < k, zeros for the padded rows k <= r < k_pad (the
// weight-gradient GEMMs multiply them by zero gradients: they must hold finite values).  One workgroup per (row, plane).
__global__ __launch_bounds__(64) void k_deform_compact_rows(int64_t k, int width /* floats, a multiple of 4 */, int64_t src_plane, int64_t dst_plane,
                                                            const int32_t* __restrict__ idx, const float* __restrict__ src, float* __restrict__ dst)
{
    const int64_t r = blockIdx.x;
    const int plane = blockIdx.y;
    float4* d = reinterpret_cast<float4*>(dst + plane * dst_plane + r * width);
    if (r < k) {
        const float4* s4 = reinterpret_cast<const float4*>(src + plane * src_plane + (int64_t)idx[r] * width);
        for (int i = threadIdx.x; i < width / 4; i += 64) d[i] = s4[i];
    } else {
        for (int i = threadIdx.x; i < width / 4; i += 64) d[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// upstream gradient rows (3 and 48 floats: 4-byte pieces)
__global__ __launch_bounds__(256) void k_deform_compact_grads(int64_t k, const int32_t* __restrict__ idx, const float* __restrict__ gx,
                                                              const float* __restrict__ gs, float* __restrict__ gx_c, float* __restrict__ gs_c)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = e / 51;
    if (r >= k) return;
    const int c = (int)(e - r * 51);
    const int64_t row = idx[r];
    if (c < 3) { if (gx) gx_c[r * 3 + c] = gx[row * 3 + c]; }
    else if (gs) gs_c[r * 48 + (c - 3)] = gs[row * 48 + (c - 3)];
}

// gft_deform_backward_rows: the plan of the launches behind it, from the row count a kernel left on the device
__global__ void k_deform_plan(const uint32_t* __restrict__ count, int64_t cap, DevPlan* __restrict__ plan, uint32_t* __restrict__ rows_out)
{
    int64_t k = (int64_t)*count;
    if (k > cap) k = cap;
    if (rows_out) *rows_out = (uint32_t)k;
    DevPlan p;
    p.n = k;
    p.n_ext = pad_points(k);
    p.tiles_per_split = 1;
    p.splits = 0;
    if (k > 0) p.splits = dw_splits(p.n_ext, &p.tiles_per_split);
    *plan = p;
}

}  // namespace


// encoded inputs of an (xyz_multires, t_multires) network, or -1 when the kernels do not hold it
static int arch_inputs(int xm, int tm)
{
    if (xm < 0 || tm < 0 || xm > 16 || tm > 24) return -1;
    const int in = 3 + 6 * xm + 1 + 2 * tm;
    return in <= DF_INK ? in : -1;
}

extern "C" int gft_deform_inputs(int xyz_multires, int t_multires) { return arch_inputs(xyz_multires, t_multires); }

extern "C" size_t gft_deform_packed_bytes(void) { return (size_t)(DF_PACKED_FLOATS + DF_BF_FLOATS + DF_H_FLOATS + DF_FLAG_FLOATS) * sizeof(float); }

extern "C" size_t gft_deform_saved_bytes(int64_t n)
{
    if (n <= 0) return 0;
    return (size_t)pad_points(n) * (DF_EMB + DF_D * DF_W + DF_D * DF_SIGN_WORDS) * sizeof(float);
}

extern "C" size_t gft_deform_scratch_bytes(int64_t n)
{
    if (n <= 0) return 0;
    const int64_t n_pad = pad_points(n);
    int tps;
    const int splits = dw_splits(n_pad, &tps);
    // dz | dzh | the splits' partial sums | the points' largest |dz| per layer | the range flag of k_deform_dw_h
    return ((size_t)n_pad * (DF_D * DF_W + DF_HEAD) + (size_t)splits * DW_PART_FLOATS + (size_t)n_pad * DF_D + 4) * sizeof(float);
}

extern "C" int gft_deform_pack(void* hip_stream, int xyz_multires, int t_multires, const gft_deform_params* p, void* packed)
{
    if (!p || !packed) return gft_fail("gft_deform_pack: NULL argument");
    const int in = arch_inputs(xyz_multires, t_multires);
    if (in < 0)
        return gft_fail("gft_deform_pack: xyz_multires = %d, t_multires = %d give more than %d encoded inputs", xyz_multires,
                        t_multires, DF_INK);
    for (int l = 0; l < DF_D; l++)
        if (!p->linear_w[l] || !p->linear_b[l]) return gft_fail("gft_deform_pack: linear.%d parameters are NULL", l);
    if (!p->xyz_w || !p->xyz_b || !p->r_w || !p->r_b || !p->g_w || !p->g_b || !p->b_w || !p->b_b)
        return gft_fail("gft_deform_pack: head parameters are NULL");
    PackArgs a;
    a.p = *p;
    a.out = (float*)packed;
    a.in = in;
    a.flag = reinterpret_cast<uint32_t*>((float*)packed + DF_FLAG_OFF);
    hipLaunchKernelGGL(k_deform_pack, dim3((unsigned)((DF_PACKED_FLOATS + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, a);
    GFT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_deform_pack_bf, dim3((unsigned)((DF_BF_ELEMS + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       (const float*)packed, reinterpret_cast<__bf16*>((float*)packed + DF_PACKED_FLOATS));
    GFT_CHECK_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_deform_pack_h, dim3((unsigned)((DF_BF_ELEMS + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream,
                       (const float*)packed, reinterpret_cast<_Float16*>((float*)packed + DF_PACKED_FLOATS + DF_BF_FLOATS),
                       reinterpret_cast<uint32_t*>((float*)packed + DF_FLAG_OFF));
    GFT_CHECK_HIP(hipGetLastError());
    return 0;
}

// (plan != NULL: n is the capacity -- the launch's size and the buffers' plane stride --, the kernels take the point count from the plan)
static int deform_forward_impl(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const float* xyz, const float* t,
                               int64_t t_stride, const void* packed, void* saved, float* d_xyz, float* d_sh, const DevPlan* plan)
{
    if (n < 0) return gft_fail("gft_deform_forward: n < 0");
    if (arch_inputs(xyz_multires, t_multires) < 0)
        return gft_fail("gft_deform_forward: xyz_multires = %d, t_multires = %d give more than %d encoded inputs", xyz_multires,
                        t_multires, DF_INK);
    if (n == 0) return 0;
    if (!xyz || !t || !packed || !d_xyz || !d_sh) return gft_fail("gft_deform_forward: NULL argument");
    if (t_stride != 0 && t_stride != 1) return gft_fail("gft_deform_forward: t_stride must be 0 or 1");
    if (n > ((int64_t)1 << 31) * 16) return gft_fail("gft_deform_forward: n too large");
    GFT_CHECK_HIP(set_attrs());
    FwdArgs a;
    a.plan = plan;
    a.n = n;
    a.n_pad = pad_points(n);
    a.t_stride = t_stride;
    a.xm = xyz_multires;
    a.tm = t_multires;
    a.xyz = xyz;
    a.t = t;
    a.packed = (const float*)packed;
    a.emb = (float*)saved;
    a.acts = saved ? (float*)saved + a.n_pad * DF_EMB : nullptr;
    a.signs = saved ? reinterpret_cast<uint32_t*>(a.acts + a.n_pad * DF_D * DF_W) : nullptr;
    a.d_xyz = d_xyz;
    a.d_sh = d_sh;
    // all padded rows are computed and saved: the weight-gradient GEMMs multiply them (by zero gradients)
    const dim3 grid((unsigned)(a.n_pad / (32 * DF_NR_FWD)));
    a.gen = 0;
    a.only_if = 0;
    if (bf16_planes() && fp16_forward()) {
        static std::atomic<uint32_t> calls{0};
        do a.gen = ++calls; while (a.gen == 0);
        if (saved)
            hipLaunchKernelGGL((k_deform_fwd_h<true, DF_FWD_H_SAVE_WAVES>), grid, dim3(64 * DF_FWD_H_SAVE_WAVES), DF_FWD_H_SAVE_LDS,
                               (hipStream_t)hip_stream, a);
        else
            hipLaunchKernelGGL((k_deform_fwd_h<false, DF_FWD_WAVES>), grid, dim3(64 * DF_FWD_WAVES), DF_FWD_H_LDS, (hipStream_t)hip_stream, a);
        GFT_CHECK_HIP(hipGetLastError());
        // the fp32-range walk behind it: its workgroups return at once unless the fp16 planes could not hold a value
        a.only_if = a.gen;
        if (saved) hipLaunchKernelGGL(k_deform_fwd_bf<true>, grid, dim3(64 * DF_FWD_WAVES), DF_FWD_BF_LDS, (hipStream_t)hip_stream, a);
        else hipLaunchKernelGGL(k_deform_fwd_bf<false>, grid, dim3(64 * DF_FWD_WAVES), DF_FWD_BF_LDS, (hipStream_t)hip_stream, a);
    } else if (bf16_planes()) {
        if (saved) hipLaunchKernelGGL(k_deform_fwd_bf<true>, grid, dim3(64 * DF_FWD_WAVES), DF_FWD_BF_LDS, (hipStream_t)hip_stream, a);
        else hipLaunchKernelGGL(k_deform_fwd_bf<false>, grid, dim3(64 * DF_FWD_WAVES), DF_FWD_BF_LDS, (hipStream_t)hip_stream, a);
    } else {
        if (saved) hipLaunchKernelGGL(k_deform_fwd<true>, grid, dim3(256), DF_FWD_LDS, (hipStream_t)hip_stream, a);
        else hipLaunchKernelGGL(k_deform_fwd<false>, grid, dim3(256), DF_FWD_LDS, (hipStream_t)hip_stream, a);
    }
    GFT_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int gft_deform_forward(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const float* xyz, const float* t,
                                  int64_t t_stride, const void* packed, void* saved, float* d_xyz, float* d_sh)
{
    return deform_forward_impl(hip_stream, xyz_multires, t_multires, n, xyz, t, t_stride, packed, saved, d_xyz, d_sh, nullptr);
}

static int deform_backward_impl(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const void* packed,
                                const void* saved, const float* g_d_xyz, const float* g_d_sh, void* scratch,
                                const gft_deform_grads* g, const DevPlan* plan)
{
    if (n < 0) return gft_fail("gft_deform_backward: n < 0");
    const int DF_IN = arch_inputs(xyz_multires, t_multires);
    if (DF_IN < 0)
        return gft_fail("gft_deform_backward: xyz_multires = %d, t_multires = %d give more than %d encoded inputs", xyz_multires,
                        t_multires, DF_INK);
    if (!g) return gft_fail("gft_deform_backward: grads is NULL");
    for (int l = 0; l < DF_D; l++)
        if (!g->linear_w[l] || !g->linear_b[l]) return gft_fail("gft_deform_backward: linear.%d gradient pointers are NULL", l);
    if (!g->xyz_w || !g->xyz_b || !g->r_w || !g->r_b || !g->g_w || !g->g_b || !g->b_w || !g->b_b)
        return gft_fail("gft_deform_backward: head gradient pointers are NULL");
    hipStream_t s = (hipStream_t)hip_stream;
    if (n == 0) {
        for (int l = 0; l < DF_D; l++) {
            const size_t in = l == 0 ? DF_IN : l == 5 ? DF_W + DF_IN : DF_W;
            GFT_CHECK_HIP(gft_zero_async(g->linear_w[l], (size_t)DF_W * in * sizeof(float), s));
            GFT_CHECK_HIP(gft_zero_async(g->linear_b[l], DF_W * sizeof(float), s));
        }
        GFT_CHECK_HIP(gft_zero_async(g->xyz_w, 3 * DF_W * sizeof(float), s));
        GFT_CHECK_HIP(gft_zero_async(g->xyz_b, 3 * sizeof(float), s));
        float* hw[3] = {g->r_w, g->g_w, g->b_w};
        float* hb[3] = {g->r_b, g->g_b, g->b_b};
        for (int c = 0; c < 3; c++) {
            GFT_CHECK_HIP(gft_zero_async(hw[c], 16 * DF_W * sizeof(float), s));
            GFT_CHECK_HIP(gft_zero_async(hb[c], 16 * sizeof(float), s));
        }
        return 0;
    }
    if (!packed || !saved || !scratch) return gft_fail("gft_deform_backward: NULL argument");
    GFT_CHECK_HIP(set_attrs());
    const int64_t n_pad = pad_points(n);
    const float* emb = (const float*)saved;
    const float* acts = emb + n_pad * DF_EMB;
    float* dz = (float*)scratch;
    float* dzh = dz + n_pad * DF_D * DF_W;
    float* part = dzh + n_pad * DF_HEAD;
    int tps;
    const int splits = dw_splits(n_pad, &tps);
    float* rowmax = part + (int64_t)splits * DW_PART_FLOATS;
    uint32_t* xflag = reinterpret_cast<uint32_t*>(rowmax + n_pad * DF_D);
    const bool h_planes = bf16_planes() && fp16_backward();
    {
        BwdArgs a;
        a.plan = plan;
        a.n = n; a.n_pad = n_pad;
        a.packed = (const float*)packed;
        a.signs = reinterpret_cast<const uint32_t*>(acts + n_pad * DF_D * DF_W);
        a.g_dxyz = g_d_xyz; a.g_dsh = g_d_sh;
        a.dz = dz; a.dzh = dzh;
        a.only_if_wflag = 0;
        a.rowmax = rowmax; a.xflag = xflag;
        if (h_planes) {
            hipLaunchKernelGGL(k_deform_bwd_h, dim3((unsigned)(n_pad / 64)), dim3(64 * DF_BWD_WAVES), DF_BWD_H_LDS, s, a);
            GFT_CHECK_HIP(hipGetLastError());
            // the fp32-range walk behind it: its workgroups return at once unless a weight does not fit the fp16 planes
            a.only_if_wflag = 1;
            hipLaunchKernelGGL(k_deform_bwd_bf, dim3((unsigned)(n_pad / 64)), dim3(64 * DF_BWD_WAVES), DF_BWD_BF_LDS, s, a);
        } else if (bf16_planes()) hipLaunchKernelGGL(k_deform_bwd_bf, dim3((unsigned)(n_pad / 64)), dim3(64 * DF_BWD_WAVES), DF_BWD_BF_LDS, s, a);
        else hipLaunchKernelGGL(k_deform_bwd, dim3((unsigned)(n_pad / (32 * DF_NR_BWD))), dim3(256), DF_BWD_LDS, s, a);
        GFT_CHECK_HIP(hipGetLastError());
    }
    {
        DwArgs a;
        a.plan = plan;
        a.n_pad = n_pad; a.n_ext = n_pad;
        a.tiles_per_split = tps;
        a.splits = splits;
        a.emb = emb; a.acts = acts; a.dz = dz; a.dzh = dzh;
        a.part = part;
        a.first_block = 0;
        a.packed = (const float*)packed; a.rowmax = rowmax; a.xflag = xflag; a.only_if = 0;
        if (bf16_planes()) {
            // Two launches of the same kernel: the hidden-layer jobs (7 x splits workgroups, one wave per SIMD, 96 KB
            // of LDS), then the light jobs (dW of layer 0, of the encoding rows of layer 5, of the heads), which need no
            // LDS and get the CUs to themselves.  Measured at 300 k points: 1.21 + 0.29 ms back to back; with the
            // light jobs on a side stream beside the heavy ones 1.53 ms (the two share the CUs' issue slots and the
            // heavy waves' hand-placed schedule has no room for a guest), light first 1.54 ms.
            static std::atomic<uint64_t> done{0};
            GFT_CHECK_HIP(gft_lds_opt_in(reinterpret_cast<const void*>(&k_deform_dw_bf), DW_SH_LDS, done));
            if (h_planes) {
                // the hidden layers on two fp16 planes; the bf16 job behind it returns at once unless a weight or an
                // activation did not fit them
                static std::atomic<uint64_t> done_h2{0}, done_h3{0};
                GFT_CHECK_HIP(gft_lds_opt_in(reinterpret_cast<const void*>(&k_deform_dw_h<2>), DW_H_LDS, done_h2));
                GFT_CHECK_HIP(gft_lds_opt_in(reinterpret_cast<const void*>(&k_deform_dw_h<3>), DW_H_LDS, done_h3));
                // (three steps of loads in flight where the grid is a round or two of workgroups: see dw_job_h_shared)
                if (n_pad < 200000) hipLaunchKernelGGL(k_deform_dw_h<3>, dim3(7 * splits), dim3(256), DW_H_LDS, s, a);
                else hipLaunchKernelGGL(k_deform_dw_h<2>, dim3(7 * splits), dim3(256), DW_H_LDS, s, a);
                GFT_CHECK_HIP(hipGetLastError());
                a.only_if = 1;
            }
            hipLaunchKernelGGL(k_deform_dw_bf, dim3(7 * splits), dim3(256), DW_SH_LDS, s, a);
            GFT_CHECK_HIP(hipGetLastError());
            a.only_if = 0;
            DwArgs light = a;
            light.first_block = 7 * splits;
            hipLaunchKernelGGL(k_deform_dw_bf, dim3(3 * splits), dim3(256), 0, s, light);
        } else hipLaunchKernelGGL(k_deform_dw, dim3(DW_JOBS * splits), dim3(256), 0, s, a);
        GFT_CHECK_HIP(hipGetLastError());
    }
    {
        ReduceArgs a;
        a.plan = plan;
        a.part = part;
        a.splits = splits;
        int k = 0;
        auto seg = [&](float* dst, int rows, int cols, int dst_ld, int dst_col0, int64_t src_off, int src_ld, int mul, int add) {
            ReduceSeg& sg = a.seg[k++];
            sg.dst = dst; sg.rows = rows; sg.cols = cols; sg.dst_ld = dst_ld; sg.dst_col0 = dst_col0;
            sg.src_off = src_off; sg.src_ld = src_ld; sg.row_mul = mul; sg.row_add = add;
        };
        seg(g->linear_w[0], DF_W, DF_IN, DF_IN, 0, DW_OFF_L0, DF_EMB, 1, 0);
        for (int l = 1; l < DF_D; l++) {
            if (l == 5) {
                seg(g->linear_w[5], DF_W, DF_IN, DF_W + DF_IN, 0, DW_OFF_L5E, DF_EMB, 1, 0);
                seg(g->linear_w[5], DF_W, DF_W, DF_W + DF_IN, DF_IN, DW_OFF_L(5), DF_W, 1, 0);
            } else {
                seg(g->linear_w[l], DF_W, DF_W, DF_W, 0, DW_OFF_L(l), DF_W, 1, 0);
            }
        }
        for (int l = 0; l < DF_D; l++) seg(g->linear_b[l], 1, DF_W, DF_W, 0, DW_OFF_BIAS + l * DF_W, DF_W, 1, 0);
        float* hw[3] = {g->r_w, g->g_w, g->b_w};
        float* hb[3] = {g->r_b, g->g_b, g->b_b};
        for (int c = 0; c < 3; c++) {
            seg(hw[c], 16, DF_W, DF_W, 0, DW_OFF_HEAD, DF_W, 3, c);
            // bias: a [16] vector gathered with stride 3 from the head's 64 column sums
            seg(hb[c], 16, 1, 1, 0, DW_OFF_BIAS + DF_D * DF_W, 1, 3, c);
        }
        seg(g->xyz_w, 3, DF_W, DF_W, 0, DW_OFF_HEAD, DF_W, 1, 48);
        seg(g->xyz_b, 3, 1, 1, 0, DW_OFF_BIAS + DF_D * DF_W, 1, 1, 48);
        a.nseg = k;
        hipLaunchKernelGGL(k_deform_reduce, dim3((DF_W * DF_W + 255) / 256, k), dim3(256), 0, s, a);
        GFT_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" int gft_deform_backward(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const void* packed,
                                   const void* saved, const float* g_d_xyz, const float* g_d_sh, void* scratch,
                                   const gft_deform_grads* g)
{
    return deform_backward_impl(hip_stream, xyz_multires, t_multires, n, packed, saved, g_d_xyz, g_d_sh, scratch, g, nullptr);
}

extern "C" int gft_deform_compact(void* hip_stream, int64_t n, int64_t k, const uint8_t* mask, const int32_t* rank, const void* saved,
                                  const float* g_d_xyz, const float* g_d_sh, int32_t* idx, void* saved_c, float* g_d_xyz_c,
                                  float* g_d_sh_c)
{
    if (n < 0 || k < 0 || k > n) return gft_fail("gft_deform_compact: bad row counts");
    if (k == 0) return 0;
    if (!mask || !rank || !saved || !idx || !saved_c) return gft_fail("gft_deform_compact: NULL argument");
    if ((g_d_xyz && !g_d_xyz_c) || (g_d_sh && !g_d_sh_c)) return gft_fail("gft_deform_compact: a gradient has no destination");
    hipStream_t s = (hipStream_t)hip_stream;
    const int64_t n_pad = pad_points(n), k_pad = pad_points(k);
    hipLaunchKernelGGL(k_deform_compact_index, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, mask, rank, idx);
    GFT_CHECK_HIP(hipGetLastError());
    const float* src = (const float*)saved;
    float* dst = (float*)saved_c;
    // encoding [n_pad, E] | activations [8, n_pad, 256] | ReLU sign words [8, n_pad, 8] (gft_deform_saved_bytes)
    hipLaunchKernelGGL(k_deform_compact_rows, dim3((unsigned)k_pad, 1), dim3(64), 0, s, k, DF_EMB, n_pad * DF_EMB, k_pad * DF_EMB, idx, src, dst);
    src += n_pad * DF_EMB; dst += k_pad * DF_EMB;
    hipLaunchKernelGGL(k_deform_compact_rows, dim3((unsigned)k_pad, DF_D), dim3(64), 0, s, k, DF_W, n_pad * DF_W, k_pad * DF_W, idx, src, dst);
    src += n_pad * DF_D * DF_W; dst += k_pad * DF_D * DF_W;
    hipLaunchKernelGGL(k_deform_compact_rows, dim3((unsigned)k_pad, DF_D), dim3(64), 0, s, k, DF_SIGN_WORDS, n_pad * DF_SIGN_WORDS,
                       k_pad * DF_SIGN_WORDS, idx, src, dst);
    if (g_d_xyz || g_d_sh)
        hipLaunchKernelGGL(k_deform_compact_grads, dim3((unsigned)((k * 51 + 255) / 256)), dim3(256), 0, s, k, idx, g_d_xyz, g_d_sh, g_d_xyz_c, g_d_sh_c);
    GFT_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- backward over the rows with an upstream gradient, counted on the device (nothing is read back: capturable) ----------
namespace {
struct RowsWork {      // byte offsets into the caller's work buffer (all multiples of 256)
    size_t plan, count, rank_scratch, mask, rank, xs, ts, gx, gs, dx, ds, saved, scratch, total;
};
RowsWork rows_work(int64_t n)
{
    RowsWork w;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    w.plan = take(sizeof(DevPlan));
    w.count = take(sizeof(uint32_t));
    w.rank_scratch = take(gft_rows_rank_scratch_bytes(n));
    w.mask = take((size_t)n);
    w.rank = take((size_t)n * 4);
    w.xs = take((size_t)n * 12);
    w.ts = take((size_t)n * 4);
    w.gx = take((size_t)n * 12);
    w.gs = take((size_t)n * 192);
    w.dx = take((size_t)n * 12);
    w.ds = take((size_t)n * 192);
    w.saved = take(gft_deform_saved_bytes(n));
    w.scratch = take(gft_deform_scratch_bytes(n));
    w.total = o;
    return w;
}
}  // namespace

extern "C" size_t gft_deform_rows_work_bytes(int64_t n) { return n <= 0 ? 0 : rows_work(n).total; }

extern "C" int gft_deform_backward_rows(void* hip_stream, int xyz_multires, int t_multires, int64_t n, const void* packed,
                                        const float* xyz, const float* t, int64_t t_stride, const float* g_d_xyz,
                                        const float* g_d_sh, void* work, const gft_deform_grads* g, uint32_t* rows_out)
{
    if (n < 0) return gft_fail("gft_deform_backward_rows: n < 0");
    if (n == 0 || (!g_d_xyz && !g_d_sh)) {      // no row has a gradient: the dense entry's zero fill
        if (rows_out) GFT_CHECK_HIP(gft_zero_async(rows_out, sizeof(uint32_t), (hipStream_t)hip_stream));
        return deform_backward_impl(hip_stream, xyz_multires, t_multires, 0, packed, nullptr, nullptr, nullptr, nullptr, g, nullptr);
    }
    if (n > 0x7fffffffll) return gft_fail("gft_deform_backward_rows: n too large");
    if (!packed || !xyz || !t || !work) return gft_fail("gft_deform_backward_rows: NULL argument");
    if (t_stride != 0 && t_stride != 1) return gft_fail("gft_deform_backward_rows: t_stride must be 0 or 1");
    if (((uintptr_t)work & 255) != 0) return gft_fail("gft_deform_backward_rows: work must be 256-byte aligned");
    hipStream_t s = (hipStream_t)hip_stream;
    const RowsWork w = rows_work(n);
    char* base = (char*)work;
    DevPlan* plan = reinterpret_cast<DevPlan*>(base + w.plan);
    uint32_t* count = reinterpret_cast<uint32_t*>(base + w.count);
    uint8_t* mask = reinterpret_cast<uint8_t*>(base + w.mask);
    int32_t* rank = reinterpret_cast<int32_t*>(base + w.rank);
    float* xs = reinterpret_cast<float*>(base + w.xs);
    float* ts = reinterpret_cast<float*>(base + w.ts);
    float* gx = reinterpret_cast<float*>(base + w.gx);
    float* gs = reinterpret_cast<float*>(base + w.gs);
    // which rows, their output rows, how many (on the device), the plan of everything behind
    if (gft_rows_any_nonzero(s, n, g_d_xyz ? 3 : 0, g_d_xyz, g_d_sh ? 48 : 0, g_d_sh, mask)) return 1;
    if (gft_rows_rank_dev(s, n, mask, rank, base + w.rank_scratch, count)) return 1;
    hipLaunchKernelGGL(k_deform_plan, dim3(1), dim3(1), 0, s, count, n, plan, rows_out);
    GFT_CHECK_HIP(hipGetLastError());
    // the selected rows' inputs and upstream gradients, in row order (dst[rank[i]] = src[i])
    if (gft_rows_gather(s, n, mask, rank, xyz, xs, 12)) return 1;
    if (t_stride == 1 && gft_rows_gather(s, n, mask, rank, t, ts, 4)) return 1;
    if (g_d_xyz && gft_rows_gather(s, n, mask, rank, g_d_xyz, gx, 12)) return 1;
    if (g_d_sh && gft_rows_gather(s, n, mask, rank, g_d_sh, gs, 192)) return 1;
    // their activations from a saving forward over them alone (a point's activations do not depend on its batch), then the
    // backward over them; launches of the capacity, extents from the plan
    if (deform_forward_impl(s, xyz_multires, t_multires, n, xs, t_stride == 1 ? ts : t, t_stride, packed, base + w.saved,
                            reinterpret_cast<float*>(base + w.dx), reinterpret_cast<float*>(base + w.ds), plan))
        return 1;
    return deform_backward_impl(s, xyz_multires, t_multires, n, packed, base + w.saved, g_d_xyz ? gx : nullptr, g_d_sh ? gs : nullptr,
                                base + w.scratch, g, plan);
}
