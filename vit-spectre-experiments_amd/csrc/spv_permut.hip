// spv_permut.hip -- MHPermutMix gather (signed multi-head permutation) forward and backward.
//
// forward : g[b, h, j] = x[b, perm[h][j]] * sign[h][j]
// backward: dx[b, i]   = sum_h sign[h][inv_h(i)] * dg[b, h, inv_h(i)]     (each perm[h] is a bijection)
// Both directions are GATHERS through a packed uint32 table (bit 31 = sign, bits 0..30 = source index):
// `fwd` table idx[h][j] = perm[h][j], `inv` table idx[h][i] = inv_h(i).  No atomics, deterministic.
//
// One workgroup per sample stages the source row (forward: the d-element token row; backward: one head's
// d-element slice at a time) in LDS with coalesced 16-byte loads, then gathers from LDS (random 2/4-byte
// LDS reads instead of random global reads) and writes the destination coalesced.  Rows too large for the
// 160 KB LDS (Base/224: d = 151 296) gather straight from global memory (L2 resident per sample).
//
// Every workgroup walks the whole table of its direction, so at bs 512 the uint32 table (2.1 MB at the Small width) costs
// 1.1 GB of L2 -> CU traffic per launch -- twice the bytes the gather itself writes.  When d <= 65 536 the pack step therefore
// also emits a COMPACT form (16-bit indices + one sign bit per element, 2.125 bytes per element) and the bf16 kernels use it;
// the forward kernel additionally serves TWO samples per workgroup from one pass over the table (their rows interleaved in LDS
// as bf16 pairs, so one 4-byte LDS read fetches both), which halves the table traffic again.
#include "spv_common.h"

namespace {

constexpr int PT = 1024;         // threads per workgroup
constexpr int LDS_LIMIT = 150 * 1024;

// compact tables exist iff every index fits 16 bits and head slices start on a sign-byte boundary
inline bool compact_ok(int d) { return d <= 65536 && d % 8 == 0; }
struct Compact {
    const uint16_t *fwd16, *inv16;
    const uint8_t *fwd_sg, *inv_sg;
};
inline Compact compact_of(const uint32_t* idx, int64_t total) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(idx + 2 * total);
    Compact c;
    c.fwd16 = reinterpret_cast<const uint16_t*>(base);
    c.inv16 = c.fwd16 + total;
    c.fwd_sg = base + 4 * total;
    c.inv_sg = c.fwd_sg + total / 8;
    return c;
}

// compact tables from the wide ones: thread per 8 elements
__global__ __launch_bounds__(256) void permut_compact_kernel(const uint32_t* __restrict__ wide, uint16_t* __restrict__ c16,
                                                             uint8_t* __restrict__ sg, int64_t octets) {
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < octets; o += (int64_t)gridDim.x * blockDim.x) {
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t w = wide[o * 8 + k];
            c16[o * 8 + k] = (uint16_t)(w & 0xffffu);
            bits |= (w >> 31) << k;
        }
        sg[o] = (uint8_t)bits;
    }
}

__global__ __launch_bounds__(256) void permut_pack_kernel(const int64_t* __restrict__ perms, const float* __restrict__ signs,
                                                          uint32_t* __restrict__ fwd, uint32_t* __restrict__ inv, int heads, int d) {
    const int64_t total = (int64_t)heads * d;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (int64_t)gridDim.x * blockDim.x) {
        const int h = (int)(f / d), j = (int)(f % d);
        const uint32_t p = (uint32_t)perms[f];
        const uint32_t s = signs[f] < 0.0f ? 0x80000000u : 0u;
        fwd[f] = p | s;
        inv[(int64_t)h * d + p] = (uint32_t)j | s;
    }
}

template <typename T> struct raw;
template <> struct raw<float> { using type = uint32_t; static constexpr uint32_t SIGN = 0x80000000u; };
template <> struct raw<bf16_t> { using type = uint16_t; static constexpr uint32_t SIGN = 0x8000u; };

// forward, LDS path: grid = batch
template <typename T>
__global__ __launch_bounds__(PT) void gather_fwd_lds_kernel(const T* __restrict__ x, const uint32_t* __restrict__ idx,
                                                            T* __restrict__ g, int heads, int d, T* __restrict__ pooled, int pw) {
    using R = typename raw<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R* row = reinterpret_cast<R*>(smem);
    const int b = blockIdx.x;
    const R* xr = reinterpret_cast<const R*>(x) + (size_t)b * d;
    constexpr int VE = 16 / sizeof(T);  // elements per 16 B
    const int nv = d / VE;
    for (int v = threadIdx.x; v < nv; v += PT) reinterpret_cast<uint4*>(row)[v] = reinterpret_cast<const uint4*>(xr)[v];
    for (int e = nv * VE + threadIdx.x; e < d; e += PT) row[e] = xr[e];
    __syncthreads();
    const int64_t total = (int64_t)heads * d;
    R* go = reinterpret_cast<R*>(g) + (size_t)b * total;
    const int64_t t4 = total >> 2;
    for (int64_t q = threadIdx.x; q < t4; q += PT) {
        const uint4 id = reinterpret_cast<const uint4*>(idx)[q];
        R o[4];
        o[0] = row[id.x & 0x7fffffffu] ^ (R)((id.x >> 31) * raw<T>::SIGN);
        o[1] = row[id.y & 0x7fffffffu] ^ (R)((id.y >> 31) * raw<T>::SIGN);
        o[2] = row[id.z & 0x7fffffffu] ^ (R)((id.z >> 31) * raw<T>::SIGN);
        o[3] = row[id.w & 0x7fffffffu] ^ (R)((id.w >> 31) * raw<T>::SIGN);
        if constexpr (sizeof(T) == 2) {
            uint2 w;
            w.x = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
            w.y = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            reinterpret_cast<uint2*>(go)[q] = w;
        } else {
            reinterpret_cast<uint4*>(go)[q] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (pooled != nullptr) {
            // the SpectreLinear skip averages pw consecutive gathered elements (layers.py:93,101): this lane has 4 of them,
            // pw / 4 neighbouring lanes (consecutive q) the rest -> DPP quad / row reduction, first lane of the group writes
            float sm;
            if constexpr (sizeof(T) == 2)
                sm = (bf2f(o[0]) + bf2f(o[1])) + (bf2f(o[2]) + bf2f(o[3]));
            else
                sm = (__uint_as_float(o[0]) + __uint_as_float(o[1])) + (__uint_as_float(o[2]) + __uint_as_float(o[3]));
            const int gl = pw >> 2;  // lanes per window: 1, 2, 4 or 8 (host checks)
            if (gl >= 2) sm += dpp_mov<0xB1>(sm);   // quad_perm [1,0,3,2]
            if (gl >= 4) sm += dpp_mov<0x4E>(sm);   // quad_perm [2,3,0,1]
            if (gl >= 8) sm += dpp_mov<0x124>(sm);  // row_ror:4 -> lanes 0-3 also hold lanes 4-7's sum
            if ((q & (gl - 1)) == 0) io<T>::st(pooled + (size_t)b * (total / pw) + q / gl, sm * (1.0f / (float)pw));
        }
    }
    for (int64_t f = (t4 << 2) + threadIdx.x; f < total; f += PT) {
        const uint32_t id = idx[f];
        go[f] = row[id & 0x7fffffffu] ^ (R)((id >> 31) * raw<T>::SIGN);
    }
}

// forward, global path: grid = (chunks, batch)
template <typename T>
__global__ __launch_bounds__(256) void gather_fwd_global_kernel(const T* __restrict__ x, const uint32_t* __restrict__ idx,
                                                                T* __restrict__ g, int heads, int d) {
    using R = typename raw<T>::type;
    const int b = blockIdx.y;
    const R* xr = reinterpret_cast<const R*>(x) + (size_t)b * d;
    const int64_t total = (int64_t)heads * d;
    R* go = reinterpret_cast<R*>(g) + (size_t)b * total;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t id = idx[f];
        go[f] = xr[id & 0x7fffffffu] ^ (R)((id >> 31) * raw<T>::SIGN);
    }
}

// Rows that do not fit the LDS (Spectre-ViT-Base at 224 / 16: d = 197 x 768 = 151 296 bf16 = 296 KB) in PARTS that do: a workgroup owns
// PARTS_CH gathered elements, keeps their table entries in registers, and for each part of the sample's row -- staged in LDS
// by coalesced 16-byte loads -- picks up the elements whose source lies in it.  The row is read from L2 once per workgroup instead of
// being hit by one random two-byte global load per element (gather_fwd_global_kernel: 592 us per Base layer at bs 64, 12 of the
// student step's 30 ms together with the backward below).  grid = (chunks of PARTS_CH elements, batch).
constexpr int PARTS_T = 512;     // threads: 2 waves per SIMD, so that 64 table entries + values per thread stay in registers
constexpr int PARTS_G = 8;       // groups of 8 CONSECUTIVE elements per thread (two 16-byte table loads, one 16-byte store per group)
constexpr int PARTS_CH = PARTS_T * PARTS_G * 8;   // elements per workgroup: 32 768
typedef unsigned gp_u32x4 __attribute__((ext_vector_type(4)));
constexpr int PARTS_GF = 8;      // the forward: 8 groups per thread too (12 and 16 spill: 48 / 168 VGPRs)
constexpr int PARTS_CHF = PARTS_T * PARTS_GF * 8;
__global__ __launch_bounds__(PARTS_T) void gather_fwd_parts_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ idx,
                                                                   uint16_t* __restrict__ g, int heads, int d, int part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gp_smem[];
    const uint16_t* row = reinterpret_cast<const uint16_t*>(gp_smem);
    const int b = blockIdx.y;
    const int total = heads * d;   // < 2^31 (host), a multiple of 8
    const int f0 = blockIdx.x * PARTS_CHF + threadIdx.x * 8;   // group gi starts at f0 + gi * PARTS_T * 8
    const uint16_t* xr = x + (size_t)b * d;
    gp_u32x4 id[PARTS_GF][2], out[PARTS_GF];
#pragma unroll
    for (int gi = 0; gi < PARTS_GF; ++gi) {
        const int f = f0 + gi * PARTS_T * 8;
        const bool in = f < total;
        id[gi][0] = in ? *reinterpret_cast<const gp_u32x4*>(idx + f) : gp_u32x4{~0u, ~0u, ~0u, ~0u};   // (an index that lies in no part)
        id[gi][1] = in ? *reinterpret_cast<const gp_u32x4*>(idx + f + 4) : gp_u32x4{~0u, ~0u, ~0u, ~0u};
        out[gi] = gp_u32x4{0u, 0u, 0u, 0u};
    }
    for (int base = 0; base < d; base += part) {
        const int len = min(part, d - base);   // multiples of 8 (host)
        for (int p = threadIdx.x * 8; p < len; p += PARTS_T * 8)
            *reinterpret_cast<gp_u32x4*>(gp_smem + (size_t)p * 2) = *reinterpret_cast<const gp_u32x4*>(xr + base + p);
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < PARTS_GF; ++gi)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t w = id[gi][e >> 2][e & 3];
                const uint32_t pos = (w & 0x7fffffffu) - (uint32_t)base;
                if (pos < (uint32_t)len) {
                    const uint32_t v = (uint32_t)row[pos] ^ ((w >> 31) << 15);
                    out[gi][e >> 1] |= v << (16 * (e & 1));
                }
            }
        __syncthreads();
    }
    uint16_t* go = g + (size_t)b * total;
#pragma unroll
    for (int gi = 0; gi < PARTS_GF; ++gi) {
        const int f = f0 + gi * PARTS_T * 8;
        if (f < total) *reinterpret_cast<gp_u32x4*>(go + f) = out[gi];
    }
}
// backward: dx[b][i] = sum_h +-dg[b][h][inv_h(i)], heads in ascending order in fp32 (the order of gather_bwd_global_kernel): a workgroup
// owns PARTS_CH inputs and walks (head, part of that head's gradient row).  grid = (chunks of inputs, batch).
__global__ __launch_bounds__(PARTS_T) void gather_bwd_parts_kernel(const bf16_t* __restrict__ dg, const uint32_t* __restrict__ inv,
                                                                   bf16_t* __restrict__ dx, int heads, int d, int part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gpb_smem[];
    const bf16_t* row = reinterpret_cast<const bf16_t*>(gpb_smem);
    const int b = blockIdx.y;
    const int i0 = blockIdx.x * PARTS_CH + threadIdx.x * 8;
    float acc[PARTS_G][8];
#pragma unroll
    for (int gi = 0; gi < PARTS_G; ++gi)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[gi][e] = 0.0f;
    for (int h = 0; h < heads; ++h) {
        gp_u32x4 id[PARTS_G][2];
#pragma unroll
        for (int gi = 0; gi < PARTS_G; ++gi) {
            const int i = i0 + gi * PARTS_T * 8;
            const bool in = i < d;
            id[gi][0] = in ? *reinterpret_cast<const gp_u32x4*>(inv + (size_t)h * d + i) : gp_u32x4{~0u, ~0u, ~0u, ~0u};
            id[gi][1] = in ? *reinterpret_cast<const gp_u32x4*>(inv + (size_t)h * d + i + 4) : gp_u32x4{~0u, ~0u, ~0u, ~0u};
        }
        const bf16_t* src = dg + ((size_t)b * heads + h) * d;
        for (int base = 0; base < d; base += part) {
            const int len = min(part, d - base);
            for (int p = threadIdx.x * 8; p < len; p += PARTS_T * 8)
                *reinterpret_cast<gp_u32x4*>(gpb_smem + (size_t)p * 2) = *reinterpret_cast<const gp_u32x4*>(src + base + p);
            __syncthreads();
#pragma unroll
            for (int gi = 0; gi < PARTS_G; ++gi)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t w = id[gi][e >> 2][e & 3];
                    const uint32_t pos = (w & 0x7fffffffu) - (uint32_t)base;
                    if (pos < (uint32_t)len) {
                        const float v = bf2f(row[pos]);
                        acc[gi][e] += (w >> 31) ? -v : v;
                    }
                }
            __syncthreads();
        }
    }
#pragma unroll
    for (int gi = 0; gi < PARTS_G; ++gi) {
        const int i = i0 + gi * PARTS_T * 8;
        if (i < d) {
            gp_u32x4 o;
#pragma unroll
            for (int u = 0; u < 4; ++u) o[u] = pack_bf16x2(acc[gi][2 * u], acc[gi][2 * u + 1]);
            *reinterpret_cast<gp_u32x4*>(dx + (size_t)b * d + i) = o;
        }
    }
}
// ---- backward for rows longer than the LDS, as a SCATTER into output parts (round 3).  dx[b][perm_h(j)] += +-dg[b][h][j]: a workgroup
// owns one part of a sample's dx as fp32 accumulators in LDS (d / 4 of them at Base / 224) and, head after head, adds the sources whose
// target lies in its part -- listed at pack time per (head, part) in ascending source order, so the gradient row is read in (sparse)
// sequence instead of by one random two-byte load per element, and every source is read exactly once per launch.  Within a head the
// targets are distinct (a permutation): no atomics; heads are separated by a barrier and added in ascending order, as everywhere else.
// Table tail (uint32 words behind the wide + compact tables), two sets -- [0] from the forward table for this backward, [1] from the
// inverse table for the forward below --, each: off [heads][nparts + 1], cur [heads][nparts] (pack-time cursors),
// ej [heads][d] = source | sign << 31, et [heads][d] uint16 = target - part * L.
inline bool long_row(int d) { return (size_t)d * 2 > (size_t)LDS_LIMIT; }
inline int scat_parts(int d) { return (int)(((size_t)d * 4 + LDS_LIMIT - 1) / LDS_LIMIT); }
inline int scat_len(int d) { const int n = scat_parts(d); return (((d + n - 1) / n) + 7) / 8 * 8; }
inline int64_t scat_set_words(int heads, int d) {   // one list set
    const int64_t total = (int64_t)heads * d;
    return (int64_t)heads * (scat_parts(d) + 1) + (int64_t)heads * scat_parts(d) + total + (total + 1) / 2;
}
inline int64_t scat_words(int heads, int d) { return 2 * scat_set_words(heads, d); }   // [0] the backward's (from the forward table), [1] the forward's
struct ScatTail {
    int* off;
    int* cur;
    uint32_t* ej;
    uint16_t* et;
};
inline ScatTail scat_of(uint32_t* tail, int heads, int d) {
    ScatTail t;
    const int np = scat_parts(d);
    t.off = reinterpret_cast<int*>(tail);
    t.cur = t.off + (size_t)heads * (np + 1);
    t.ej = reinterpret_cast<uint32_t*>(t.cur + (size_t)heads * np);
    t.et = reinterpret_cast<uint16_t*>(t.ej + (size_t)heads * d);
    return t;
}
__global__ __launch_bounds__(256) void scat_count_kernel(const uint32_t* __restrict__ fwd, int* __restrict__ cur, int heads, int d, int L, int np) {
    // one wave per 64 consecutive sources of a head, one atomic per (wave, part) (a lane each: 1.8 M atomics on 48 counters took 5 ms)
    const int lane = threadIdx.x & 63;
    const int waves_per_head = (d + 63) / 64;
    const int64_t nw = (int64_t)heads * waves_per_head;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (int64_t)gridDim.x * 4) {
        const int h = (int)(w / waves_per_head), j = (int)(w % waves_per_head) * 64 + lane;
        const int q = j < d ? (int)((fwd[(size_t)h * d + j] & 0x7fffffffu) / (unsigned)L) : -1;
        for (int p = 0; p < np; ++p) {
            const unsigned long long m = __ballot(q == p);
            if (m != 0ull && lane == __ffsll((long long)m) - 1) atomicAdd(&cur[h * np + p], __popcll(m));
        }
    }
}
__global__ void scat_prefix_kernel(int* __restrict__ off, int* __restrict__ cur, int heads, int np) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= heads) return;
    int run = 0;
    for (int q = 0; q < np; ++q) {
        off[h * (np + 1) + q] = run;
        const int c = cur[h * np + q];
        cur[h * np + q] = run;   // the placement's cursor
        run += c;
    }
    off[h * (np + 1) + np] = run;
}
// one wave per 64 consecutive sources of a head: each part's lanes take consecutive slots (ascending inside the wave; waves of a head
// land in the order their atomics arrive -- any order gives the same sums, ascending-ish keeps the backward's reads sequential)
__global__ __launch_bounds__(256) void scat_place_kernel(const uint32_t* __restrict__ fwd, int* __restrict__ cur, uint32_t* __restrict__ ej,
                                                         uint16_t* __restrict__ et, int heads, int d, int L, int np) {
    const int lane = threadIdx.x & 63;
    const int waves_per_head = (d + 63) / 64;
    const int64_t nw = (int64_t)heads * waves_per_head;
    for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nw; w += (int64_t)gridDim.x * 4) {
        const int h = (int)(w / waves_per_head), j = (int)(w % waves_per_head) * 64 + lane;
        const bool in = j < d;
        const uint32_t e = in ? fwd[(size_t)h * d + j] : 0u;
        const unsigned tgt = e & 0x7fffffffu;
        const int q = in ? (int)(tgt / (unsigned)L) : -1;
        for (int p = 0; p < np; ++p) {
            const unsigned long long m = __ballot(q == p);
            if (m == 0ull) continue;
            int base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&cur[h * np + p], __popcll(m));
            base = __shfl(base, __ffsll((long long)m) - 1);
            if (q == p) {
                const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                ej[(size_t)h * d + slot] = (uint32_t)j | (e & 0x80000000u);
                et[(size_t)h * d + slot] = (uint16_t)(tgt - (unsigned)p * (unsigned)L);
            }
        }
    }
}
__global__ __launch_bounds__(1024) void gather_bwd_scatter_kernel(const bf16_t* __restrict__ dg, const int* __restrict__ off, const uint32_t* __restrict__ ej,
                                                                  const uint16_t* __restrict__ et, bf16_t* __restrict__ dx, int heads, int d, int L,
                                                                  int np) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gs_smem[];
    float* acc = reinterpret_cast<float*>(gs_smem);
    const int q = blockIdx.x, b = blockIdx.y;
    const int len = min(L, d - q * L);
    for (int t = threadIdx.x; t < len; t += 1024) acc[t] = 0.0f;
    __syncthreads();
    for (int h = 0; h < heads; ++h) {
        const int e0 = off[h * (np + 1) + q], e1 = off[h * (np + 1) + q + 1];
        const bf16_t* src = dg + ((size_t)b * heads + h) * d;
        const uint32_t* ejh = ej + (size_t)h * d;
        const uint16_t* eth = et + (size_t)h * d;
        int e = e0 + threadIdx.x;
        for (; e + 3 * 1024 < e1; e += 4 * 1024) {   // four independent (entry, source) load pairs in flight per thread
            uint32_t w[4];
            unsigned t[4];
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { w[u] = ejh[e + u * 1024]; t[u] = eth[e + u * 1024]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = bf2f(src[w[u] & 0x7fffffffu]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t[u]] += (w[u] >> 31) ? -v[u] : v[u];
        }
        for (; e < e1; e += 1024) {
            const uint32_t w = ejh[e];
            const float v = bf2f(src[w & 0x7fffffffu]);
            acc[eth[e]] += (w >> 31) ? -v : v;
        }
        __syncthreads();
    }
    bf16_t* o = dx + (size_t)b * d + (size_t)q * L;
    for (int t = threadIdx.x * 2; t < len; t += 2048) {   // len is even (multiple of 8)
        *reinterpret_cast<unsigned*>(o + t) = pack_bf16x2(acc[t], acc[t + 1]);
    }
}

// the forward the same way round: g[b][h][inv_h(i)] = +-x[b][i].  A workgroup owns one quarter of ONE head's output row in LDS (bf16,
// 74 KB: two workgroups per CU), fills it from the (head, quarter) list of the INVERSE table -- the sample's row is read in sparse
// sequence, every slot is written exactly once -- and stores it in whole 16-byte pieces.  grid = (parts, heads, batch).
__global__ __launch_bounds__(1024) void gather_fwd_scatter_kernel(const uint16_t* __restrict__ x, const int* __restrict__ off, const uint32_t* __restrict__ ej,
                                                                  const uint16_t* __restrict__ et, uint16_t* __restrict__ g, int heads, int d, int L,
                                                                  int np, bf16_t* __restrict__ pooled, int pw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gfs_smem[];
    uint16_t* out = reinterpret_cast<uint16_t*>(gfs_smem);
    const int q = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = min(L, d - q * L);
    const int e0 = off[h * (np + 1) + q], e1 = off[h * (np + 1) + q + 1];
    const uint16_t* src = x + (size_t)b * d;
    const uint32_t* ejh = ej + (size_t)h * d;
    const uint16_t* eth = et + (size_t)h * d;
    int e = e0 + threadIdx.x;
    for (; e + 3 * 1024 < e1; e += 4 * 1024) {
        uint32_t w[4];
        unsigned t[4];
        uint16_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { w[u] = ejh[e + u * 1024]; t[u] = eth[e + u * 1024]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[w[u] & 0x7fffffffu];
#pragma unroll
        for (int u = 0; u < 4; ++u) out[t[u]] = v[u] ^ (uint16_t)((w[u] >> 31) << 15);
    }
    for (; e < e1; e += 1024) {
        const uint32_t w = ejh[e];
        out[eth[e]] = src[w & 0x7fffffffu] ^ (uint16_t)((w >> 31) << 15);
    }
    __syncthreads();
    uint16_t* o = g + ((size_t)b * heads + h) * d + (size_t)q * L;
    for (int t = threadIdx.x * 8; t < len; t += 1024 * 8) *reinterpret_cast<uint4*>(o + t) = *reinterpret_cast<const uint4*>(out + t);
    if (pooled != nullptr) {
        // the SpectreLinear skip of the mix linear (AdaptiveAvgPool1d over every pw consecutive gathered elements; d and L are multiples of
        // pw, so no window crosses this quarter): emitted here so that the tail does not re-read the gathered tensor (232 MB per Base layer)
        bf16_t* po = pooled + (((size_t)b * heads + h) * d + (size_t)q * L) / pw;
        const float inv = 1.0f / (float)pw;
        for (int w = threadIdx.x; w < len / pw; w += 1024) {
            float a = 0.0f;
            for (int u = 0; u < pw; ++u) a += bf2f(out[w * pw + u]);
            po[w] = f2bf(a * inv);
        }
    }
}

// part length for a row of d elements: the fewest equal parts that fit the LDS, a multiple of 8 elements
inline int parts_len(int d) {
    const int cap = LDS_LIMIT / 2;
    const int n = (d + cap - 1) / cap;
    return (((d + n - 1) / n) + 7) / 8 * 8;
}

// backward, LDS path: grid = batch; thread owns elements i = (it*PT + tid)*4 .. +3, it < MAX_IT
constexpr int MAX_IT = 10;  // d <= 4 * PT * MAX_IT = 40 960
template <typename T>
__global__ __launch_bounds__(PT) void gather_bwd_lds_kernel(const T* __restrict__ dg, const uint32_t* __restrict__ inv,
                                                            T* __restrict__ dx, int heads, int d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* slice = reinterpret_cast<T*>(smem);
    const int b = blockIdx.x;
    constexpr int VE = 16 / sizeof(T);
    const int nv = d / VE;
    float acc[MAX_IT][4];
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[it][k] = 0.0f;
    for (int h = 0; h < heads; ++h) {
        const T* src = dg + ((size_t)b * heads + h) * d;
        __syncthreads();  // previous head's gathers are done before the slice is overwritten
        for (int v = threadIdx.x; v < nv; v += PT) reinterpret_cast<uint4*>(slice)[v] = reinterpret_cast<const uint4*>(src)[v];
        for (int e = nv * VE + threadIdx.x; e < d; e += PT) slice[e] = src[e];
        __syncthreads();
        const uint32_t* ih = inv + (size_t)h * d;
#pragma unroll
        for (int it = 0; it < MAX_IT; ++it) {
            const int i0 = (it * PT + threadIdx.x) * 4;
            if (i0 + 3 < d) {
                const uint4 id = *reinterpret_cast<const uint4*>(ih + i0);
                float v0 = io<T>::ld(slice + (id.x & 0x7fffffffu)), v1 = io<T>::ld(slice + (id.y & 0x7fffffffu));
                float v2 = io<T>::ld(slice + (id.z & 0x7fffffffu)), v3 = io<T>::ld(slice + (id.w & 0x7fffffffu));
                acc[it][0] += (id.x >> 31) ? -v0 : v0;
                acc[it][1] += (id.y >> 31) ? -v1 : v1;
                acc[it][2] += (id.z >> 31) ? -v2 : v2;
                acc[it][3] += (id.w >> 31) ? -v3 : v3;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i0 + k < d) {
                        const uint32_t id = ih[i0 + k];
                        float v = io<T>::ld(slice + (id & 0x7fffffffu));
                        acc[it][k] += (id >> 31) ? -v : v;
                    }
            }
        }
    }
    T* out = dx + (size_t)b * d;
#pragma unroll
    for (int it = 0; it < MAX_IT; ++it) {
        const int i0 = (it * PT + threadIdx.x) * 4;
        if (i0 + 3 < d) io<T>::st4(out + i0, acc[it]);
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k < d) io<T>::st(out + i0 + k, acc[it][k]);
        }
    }
}

// backward, global path: grid = (chunks, batch)
template <typename T>
__global__ __launch_bounds__(256) void gather_bwd_global_kernel(const T* __restrict__ dg, const uint32_t* __restrict__ inv,
                                                                T* __restrict__ dx, int heads, int d) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d; i += gridDim.x * blockDim.x) {
        float a = 0.0f;
        for (int h = 0; h < heads; ++h) {
            const uint32_t id = inv[(size_t)h * d + i];
            float v = io<T>::ld(dg + ((size_t)b * heads + h) * d + (id & 0x7fffffffu));
            a += (id >> 31) ? -v : v;
        }
        io<T>::st(dx + (size_t)b * d + i, a);
    }
}

// forward, compact table, two samples per workgroup (bf16): grid = ceil(batch / 2).  LDS word e = (x[b0][e], x[b0 + 1][e]).
__global__ __launch_bounds__(PT) void gather_fwd_pair_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ idx16,
                                                             const uint8_t* __restrict__ sg8, uint16_t* __restrict__ g, int heads,
                                                             int d, bf16_t* __restrict__ pooled, int pw, int batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* row = reinterpret_cast<uint32_t*>(smem);
    const int b0 = blockIdx.x * 2;
    const bool two = b0 + 1 < batch;
    const uint4* x0 = reinterpret_cast<const uint4*>(x + (size_t)b0 * d);
    const uint4* x1 = reinterpret_cast<const uint4*>(x + (size_t)(two ? b0 + 1 : b0) * d);
    const int nv = d >> 3;
    for (int v = threadIdx.x; v < nv; v += PT) {
        const uint4 a = x0[v], c = x1[v];
        uint4 lo, hi;
        lo.x = __builtin_amdgcn_perm(c.x, a.x, 0x05040100u); lo.y = __builtin_amdgcn_perm(c.x, a.x, 0x07060302u);
        lo.z = __builtin_amdgcn_perm(c.y, a.y, 0x05040100u); lo.w = __builtin_amdgcn_perm(c.y, a.y, 0x07060302u);
        hi.x = __builtin_amdgcn_perm(c.z, a.z, 0x05040100u); hi.y = __builtin_amdgcn_perm(c.z, a.z, 0x07060302u);
        hi.z = __builtin_amdgcn_perm(c.w, a.w, 0x05040100u); hi.w = __builtin_amdgcn_perm(c.w, a.w, 0x07060302u);
        reinterpret_cast<uint4*>(row)[2 * v] = lo;
        reinterpret_cast<uint4*>(row)[2 * v + 1] = hi;
    }
    __syncthreads();
    const int64_t total = (int64_t)heads * d;
    const int t8 = (int)(total >> 3);
    uint4* g0 = reinterpret_cast<uint4*>(g + (size_t)b0 * total);
    uint4* g1 = reinterpret_cast<uint4*>(g + (size_t)(b0 + 1) * total);
    const int gl = pw >> 3;  // lanes per pooling window: 1, 2 or 4 (host checks)
    // FU octets per thread per trip, the next trip's table reads issued before this trip's gathers: one octet per trip is a
    // chain of L2 latency -> LDS latency -> store, 65 times over (1 workgroup per CU: 128 VGPRs to spend)
    constexpr int FU = 4;
    const uint4* tab = reinterpret_cast<const uint4*>(idx16);
    uint4 idn[FU];
    uint32_t sgn[FU];
#pragma unroll
    for (int k = 0; k < FU; ++k) {
        const int q = threadIdx.x + k * PT;
        idn[k] = make_uint4(0, 0, 0, 0); sgn[k] = 0;
        if (q < t8) { idn[k] = tab[q]; sgn[k] = sg8[q]; }
    }
    for (int q0 = threadIdx.x; q0 < t8; q0 += FU * PT) {
        uint4 idc[FU];
        uint32_t sgc[FU];
#pragma unroll
        for (int k = 0; k < FU; ++k) { idc[k] = idn[k]; sgc[k] = sgn[k]; }
#pragma unroll
        for (int k = 0; k < FU; ++k) {
            const int q = q0 + (FU + k) * PT;
            if (q < t8) { idn[k] = tab[q]; sgn[k] = sg8[q]; }
        }
#pragma unroll
      for (int u = 0; u < FU; ++u) {
        const int q = q0 + u * PT;
        if (q >= t8) break;
        const uint4 id = idc[u];
        const uint32_t sg = sgc[u];
        uint32_t w[8];
        w[0] = row[id.x & 0xffffu]; w[1] = row[id.x >> 16];
        w[2] = row[id.y & 0xffffu]; w[3] = row[id.y >> 16];
        w[4] = row[id.z & 0xffffu]; w[5] = row[id.z >> 16];
        w[6] = row[id.w & 0xffffu]; w[7] = row[id.w >> 16];
        uint32_t o0[4], o1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t m = ((sg << (15 - 2 * k)) & 0x8000u) | ((sg << (30 - 2 * k)) & 0x80000000u);
            o0[k] = __builtin_amdgcn_perm(w[2 * k + 1], w[2 * k], 0x05040100u) ^ m;
            o1[k] = __builtin_amdgcn_perm(w[2 * k + 1], w[2 * k], 0x07060302u) ^ m;
        }
        g0[q] = make_uint4(o0[0], o0[1], o0[2], o0[3]);
        if (two) g1[q] = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        if (pooled != nullptr) {
            // the SpectreLinear skip averages pw consecutive gathered elements (layers.py:93,101): this lane has 8 of them
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s0 += __uint_as_float(o0[k] << 16) + __uint_as_float(o0[k] & 0xffff0000u);
                s1 += __uint_as_float(o1[k] << 16) + __uint_as_float(o1[k] & 0xffff0000u);
            }
            if (gl >= 2) { s0 += dpp_mov<0xB1>(s0); s1 += dpp_mov<0xB1>(s1); }   // quad_perm [1,0,3,2]
            if (gl >= 4) { s0 += dpp_mov<0x4E>(s0); s1 += dpp_mov<0x4E>(s1); }   // quad_perm [2,3,0,1]
            if ((q & (gl - 1)) == 0) {
                const float inv = 1.0f / (float)pw;
                const size_t pr = (size_t)(total / pw);
                io<bf16_t>::st(pooled + (size_t)b0 * pr + q / gl, s0 * inv);
                if (two) io<bf16_t>::st(pooled + (size_t)(b0 + 1) * pr + q / gl, s1 * inv);
            }
        }
      }
    }
}

// backward, compact table (bf16): grid = batch; thread owns elements i = (it*PT + tid)*8 .. +7, it < C_IT
constexpr int C_IT = 5;  // d <= 8 * PT * C_IT = 40 960
__global__ __launch_bounds__(PT) void gather_bwd_c16_kernel(const bf16_t* __restrict__ dg, const uint16_t* __restrict__ inv16,
                                                            const uint8_t* __restrict__ sg8, bf16_t* __restrict__ dx, int heads, int d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* slice = reinterpret_cast<uint16_t*>(smem);
    const int b = blockIdx.x;
    const int nv = d >> 3;
    float acc[C_IT][8];
#pragma unroll
    for (int it = 0; it < C_IT; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[it][k] = 0.0f;
    // (prefetching the next head's slice into registers costs 20 VGPRs, which drops the second co-resident workgroup: 133 -> 298 us)
    for (int h = 0; h < heads; ++h) {
        const uint4* src = reinterpret_cast<const uint4*>(dg + ((size_t)b * heads + h) * d);
        __syncthreads();  // previous head's gathers are done before the slice is overwritten
        for (int v = threadIdx.x; v < nv; v += PT) reinterpret_cast<uint4*>(slice)[v] = src[v];
        __syncthreads();
        const uint4* ih = reinterpret_cast<const uint4*>(inv16 + (size_t)h * d);
        const uint8_t* sh = sg8 + (((size_t)h * d) >> 3);
#pragma unroll
        for (int it = 0; it < C_IT; ++it) {
            const int o = it * PT + threadIdx.x;
            if (o < nv) {
                const uint4 id = ih[o];
                const uint32_t sg = sh[o];
                const uint32_t e[8] = {id.x & 0xffffu, id.x >> 16, id.y & 0xffffu, id.y >> 16,
                                       id.z & 0xffffu, id.z >> 16, id.w & 0xffffu, id.w >> 16};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    acc[it][k] += __uint_as_float(((uint32_t)slice[e[k]] << 16) ^ ((sg << (31 - k)) & 0x80000000u));
            }
        }
    }
    bf16_t* out = dx + (size_t)b * d;
#pragma unroll
    for (int it = 0; it < C_IT; ++it) {
        const int o = it * PT + threadIdx.x;
        if (o < nv) {
            uint4 w;
            w.x = pack_bf16x2(acc[it][0], acc[it][1]); w.y = pack_bf16x2(acc[it][2], acc[it][3]);
            w.z = pack_bf16x2(acc[it][4], acc[it][5]); w.w = pack_bf16x2(acc[it][6], acc[it][7]);
            reinterpret_cast<uint4*>(out)[o] = w;
        }
    }
}

// backward, compact table, LDS-DMA double buffer (bf16, d * 2 % 1024 == 0): ONE 1024-thread workgroup per CU owns two 66.5 KB slice
// buffers and walks (sample, head) items; the next item's slice is on its way into the other buffer (global_load_lds: no registers,
// which is what a register-staged prefetch could not afford) while this item's is gathered.  Each wave's table reads are issued
// BEFORE its DMA pieces so that the in-order memory counter lets the gather start while the DMA is still in flight.
__global__ __launch_bounds__(PT) void gather_bwd_dma_kernel(const bf16_t* __restrict__ dg, const uint16_t* __restrict__ inv16,
                                                            const uint8_t* __restrict__ sg8, bf16_t* __restrict__ dx, int heads, int d,
                                                            int batch) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int slice_bytes = d * 2;                         // multiple of 1024
    const int pieces = slice_bytes >> 10;                  // 1-KiB wave pieces per slice
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nv = d >> 3;
    const int per_wg = (batch + gridDim.x - 1) / gridDim.x;   // consecutive samples of this workgroup
    const int b_begin = blockIdx.x * per_wg, b_end = min(batch, b_begin + per_wg);
    const int items = (b_end - b_begin) * heads;
    if (items <= 0) return;
    auto dma = [&](int item, int buf) {
        const int b = b_begin + item / heads, h = item % heads;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(dg + ((size_t)b * heads + h) * d);
        for (int p = wave; p < pieces; p += PT / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (size_t)buf * slice_bytes + (size_t)p * 1024), 16, 0, 0);
    };
    float acc[C_IT][8];
#pragma unroll
    for (int it = 0; it < C_IT; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[it][k] = 0.0f;
    // Table entries travel one item ahead as well, and are issued BEHIND that item's DMA pieces: the memory counter retires in order,
    // so the wait for the entries (the copy below) is also the wait for the slice -- and nothing younger is in flight at that point.
    uint4 id_n[C_IT];
    uint32_t sg_n[C_IT];
    auto fetch_ids = [&](int item) {
        const int h = item % heads;
        const uint4* ih = reinterpret_cast<const uint4*>(inv16 + (size_t)h * d);
        const uint8_t* sh = sg8 + (((size_t)h * d) >> 3);
#pragma unroll
        for (int it = 0; it < C_IT; ++it) {
            const int o = min(it * PT + tid, nv - 1);
            id_n[it] = ih[o];
            sg_n[it] = sh[o];
        }
    };
    dma(0, 0);
    fetch_ids(0);
    for (int item = 0; item < items; ++item) {
        const int h = item % heads;
        uint4 id[C_IT];
        uint32_t sg[C_IT];
#pragma unroll
        for (int it = 0; it < C_IT; ++it) {
            id[it] = id_n[it];
            sg[it] = sg_n[it];
            asm volatile("" : "+v"(id[it].x), "+v"(id[it].y), "+v"(id[it].z), "+v"(id[it].w), "+v"(sg[it]));   // pin the wait here
        }
        __syncthreads();   // every wave's pieces of this item have landed, and every wave is done with the other buffer
        if (item + 1 < items) {
            dma(item + 1, (item + 1) & 1);
            fetch_ids(item + 1);
        }
        const uint16_t* slice = reinterpret_cast<const uint16_t*>(smem + (size_t)(item & 1) * slice_bytes);
#pragma unroll
        for (int it = 0; it < C_IT; ++it) {
            const int o = it * PT + tid;
            if (o < nv) {
                const uint32_t e[8] = {id[it].x & 0xffffu, id[it].x >> 16, id[it].y & 0xffffu, id[it].y >> 16,
                                       id[it].z & 0xffffu, id[it].z >> 16, id[it].w & 0xffffu, id[it].w >> 16};
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    acc[it][k] += __uint_as_float(((uint32_t)slice[e[k]] << 16) ^ ((sg[it] << (31 - k)) & 0x80000000u));
            }
        }
        if (h == heads - 1) {   // the sample is complete
            bf16_t* out = dx + (size_t)(b_begin + item / heads) * d;
#pragma unroll
            for (int it = 0; it < C_IT; ++it) {
                const int o = it * PT + tid;
                if (o < nv) {
                    uint4 w;
                    w.x = pack_bf16x2(acc[it][0], acc[it][1]); w.y = pack_bf16x2(acc[it][2], acc[it][3]);
                    w.z = pack_bf16x2(acc[it][4], acc[it][5]); w.w = pack_bf16x2(acc[it][6], acc[it][7]);
                    reinterpret_cast<uint4*>(out)[o] = w;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[it][k] = 0.0f;
            }
        }
    }
}

}  // namespace

// idx: spv_permut_table_words(heads, d) uint32 words: [2][heads][d] wide tables ([0] forward, [1] inverse), then -- when
// d <= 65 536 and d % 8 == 0 -- the compact tables: uint16 [2][heads][d] indices and uint8 [2][heads * d / 8] sign bits
extern "C" int spv_permut_pack(const int64_t* perms, const float* signs, uint32_t* idx, int heads, int d, void* stream) {
    SPV_CHECK(heads > 0 && d > 0, "spv_permut_pack: empty");
    SPV_CHECK((int64_t)d < (1ll << 31), "spv_permut_pack: d too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = (int64_t)heads * d;
    hipLaunchKernelGGL(permut_pack_kernel, dim3((int)std::min<int64_t>((total + 255) / 256, 2048)), dim3(256), 0, st, perms, signs,
                       idx, idx + total, heads, d);
    SPV_LAUNCH_CHECK("spv_permut_pack");
    if (compact_ok(d)) {
        const Compact c = compact_of(idx, total);
        const int64_t octets = total / 8;
        const dim3 cg((unsigned)std::min<int64_t>((octets + 255) / 256, 2048));
        hipLaunchKernelGGL(permut_compact_kernel, cg, dim3(256), 0, st, idx, const_cast<uint16_t*>(c.fwd16), const_cast<uint8_t*>(c.fwd_sg), octets);
        hipLaunchKernelGGL(permut_compact_kernel, cg, dim3(256), 0, st, idx + total, const_cast<uint16_t*>(c.inv16), const_cast<uint8_t*>(c.inv_sg), octets);
        SPV_LAUNCH_CHECK("spv_permut_pack(compact)");
    }
    if (long_row(d)) {   // the backward's scatter lists (rows longer than the LDS)
        SPV_CHECK(scat_len(d) <= 65536, "spv_permut_pack: d=%d: a part of %d targets does not fit the 16-bit target table", d, scat_len(d));
        uint32_t* tail = idx + (spv_permut_table_words(heads, d) - scat_words(heads, d));
        const int np = scat_parts(d), L = scat_len(d);
        const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
        for (int set = 0; set < 2; ++set) {   // [0]: targets of the forward table (the backward's lists); [1]: of the inverse table
            const ScatTail t = scat_of(tail + set * scat_set_words(heads, d), heads, d);
            const uint32_t* table = idx + (size_t)set * total;
            (void)hipMemsetAsync(t.cur, 0, (size_t)heads * np * sizeof(int), st);
            hipLaunchKernelGGL(scat_count_kernel, dim3(blocks), dim3(256), 0, st, table, t.cur, heads, d, L, np);
            hipLaunchKernelGGL(scat_prefix_kernel, dim3((heads + 63) / 64), dim3(64), 0, st, t.off, t.cur, heads, np);
            hipLaunchKernelGGL(scat_place_kernel, dim3(blocks), dim3(256), 0, st, table, t.cur, t.ej, t.et, heads, d, L, np);
        }
        SPV_LAUNCH_CHECK("spv_permut_pack(scatter lists)");
    }
    return 0;
}

/* 1 when spv_permut_gather_fwd can emit the pooled skip for this window on the long-row (scatter) path */
extern "C" int spv_permut_pool_supported(int heads, int d, int pool_window, int dtype) {
    (void)heads;
    return (dtype == SPV_BF16 && long_row(d) && d % 8 == 0 && pool_window > 0 && d % pool_window == 0 && scat_len(d) % pool_window == 0 &&
            !SPV_LAB_SET("SPV_GATHER_FWD_PARTS")) ? 1 : 0;
}

extern "C" int64_t spv_permut_table_words(int heads, int d) {
    const int64_t total = (int64_t)heads * d;
    return 2 * total + (compact_ok(d) ? total + (total / 4 + 3) / 4 : 0) + (long_row(d) ? scat_words(heads, d) : 0);
}

extern "C" int spv_permut_gather_fwd(const void* x, const uint32_t* idx, void* g, void* pooled, int pool_window, int batch,
                                     int heads, int d, int dtype, void* stream) {
    SPV_CHECK(batch > 0 && heads > 0 && d > 0, "spv_permut_gather_fwd: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_permut_gather_fwd: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t es = dtype == SPV_BF16 ? 2 : 4;
    const bool aligned = ((size_t)d * es) % 16 == 0 && ((int64_t)heads * d) % 4 == 0;
    const bool scat_pool = pooled != nullptr && dtype == SPV_BF16 && long_row(d) && pool_window > 0 && d % pool_window == 0 &&
                           scat_len(d) % pool_window == 0;   // the scatter path emits any window that divides its quarters
    if (pooled != nullptr && !scat_pool) {
        const int64_t total = (int64_t)heads * d;
        SPV_CHECK(aligned && (size_t)d * es <= (size_t)LDS_LIMIT, "spv_permut_gather_fwd: pooled output needs the LDS path");
        SPV_CHECK((pool_window == 4 || pool_window == 8 || pool_window == 16 || pool_window == 32) && total % pool_window == 0 &&
                      (total / 4) % PT == 0,
                  "spv_permut_gather_fwd: unsupported pool window %d", pool_window);
    }
    static const bool wide_only = SPV_LAB_SET("SPV_PERMUT_WIDE");  // A/B switch: the round-1 kernels
    const int64_t total_e = (int64_t)heads * d;
    if (!wide_only && dtype == SPV_BF16 && compact_ok(d) && (size_t)d * 4 <= (size_t)LDS_LIMIT &&
        (pooled == nullptr || ((pool_window == 8 || pool_window == 16 || pool_window == 32) && (total_e / 8) % PT == 0))) {
        const Compact c = compact_of(idx, total_e);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        SPV_COUNT_PATH(SPV_PATH_GATHER_LDS);
        hipLaunchKernelGGL(gather_fwd_pair_kernel, dim3((batch + 1) / 2), dim3(PT), (size_t)d * 4, st, (const uint16_t*)x, c.fwd16, c.fwd_sg,
                           (uint16_t*)g, heads, d, (bf16_t*)pooled, pool_window, batch);
    } else if (aligned && (size_t)d * es <= (size_t)LDS_LIMIT) {
        const size_t lds = (size_t)d * es;
        if (dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            SPV_COUNT_PATH(SPV_PATH_GATHER_LDS);
            hipLaunchKernelGGL((gather_fwd_lds_kernel<bf16_t>), dim3(batch), dim3(PT), lds, st, (const bf16_t*)x, idx, (bf16_t*)g, heads, d, (bf16_t*)pooled, pool_window);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL((gather_fwd_lds_kernel<float>), dim3(batch), dim3(PT), lds, st, (const float*)x, idx, (float*)g, heads, d, (float*)pooled, pool_window);
        }
    } else if (!SPV_LAB_SET("SPV_GATHER_FWD_PARTS") && dtype == SPV_BF16 && long_row(d) && d % 8 == 0 && (pooled == nullptr || scat_pool) &&
               (((uintptr_t)x | (uintptr_t)g) & 15) == 0 && heads <= 65535 && batch <= 65535) {
        // output quarters in LDS, filled from the inverse table's lists (spv_permut_pack)
        const ScatTail t = scat_of(const_cast<uint32_t*>(idx) + (spv_permut_table_words(heads, d) - scat_words(heads, d)) + scat_set_words(heads, d), heads, d);
        const int np = scat_parts(d), L = scat_len(d);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        hipLaunchKernelGGL(gather_fwd_scatter_kernel, dim3(np, heads, batch), dim3(1024), (size_t)L * 2, st, (const uint16_t*)x, t.off, t.ej, t.et, (uint16_t*)g, heads,
                           d, L, np, (bf16_t*)pooled, pool_window);
    } else if (dtype == SPV_BF16 && d % 8 == 0 && (((uintptr_t)x | (uintptr_t)g | (uintptr_t)idx) & 15) == 0 && (int64_t)heads * d < (1ll << 31) - PARTS_CHF) {
        // rows longer than the LDS: staged part by part (Base / 224)
        const int part = parts_len(d);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_fwd_parts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        const dim3 grid((unsigned)(((int64_t)heads * d + PARTS_CHF - 1) / PARTS_CHF), batch);
        hipLaunchKernelGGL(gather_fwd_parts_kernel, grid, dim3(PARTS_T), (size_t)part * 2, st, (const uint16_t*)x, idx, (uint16_t*)g, heads, d, part);
    } else {
        dim3 grid((unsigned)std::min<int64_t>(((int64_t)heads * d + 255) / 256, 1024), batch);
        if (dtype == SPV_BF16)
            hipLaunchKernelGGL((gather_fwd_global_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, idx, (bf16_t*)g, heads, d);
        else
            hipLaunchKernelGGL((gather_fwd_global_kernel<float>), grid, dim3(256), 0, st, (const float*)x, idx, (float*)g, heads, d);
    }
    SPV_LAUNCH_CHECK("spv_permut_gather_fwd");
    return 0;
}

extern "C" int spv_permut_gather_bwd(const void* dg, const uint32_t* idx, void* dx, int batch, int heads, int d, int dtype,
                                     void* stream) {
    SPV_CHECK(batch > 0 && heads > 0 && d > 0, "spv_permut_gather_bwd: empty");
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_permut_gather_bwd: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t* inv = idx + (size_t)heads * d;
    const size_t es = dtype == SPV_BF16 ? 2 : 4;
    const bool aligned = ((size_t)d * es) % 16 == 0 && d % 4 == 0;
    static const bool wide_only = SPV_LAB_SET("SPV_PERMUT_WIDE");
    static const bool no_dma = SPV_LAB_SET("SPV_GATHER_NO_DMA");   // A/B switch
    if (!wide_only && !no_dma && dtype == SPV_BF16 && compact_ok(d) && ((size_t)d * 2) % 1024 == 0 && (size_t)d * 4 <= (size_t)LDS_LIMIT &&
        d <= 8 * PT * C_IT && (((uintptr_t)dg | (uintptr_t)dx) & 15) == 0) {
        const Compact c = compact_of(idx, (int64_t)heads * d);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        const int wgs = std::min(batch, 256);
        hipLaunchKernelGGL(gather_bwd_dma_kernel, dim3(wgs), dim3(PT), (size_t)d * 4, st, (const bf16_t*)dg, c.inv16, c.inv_sg, (bf16_t*)dx, heads, d, batch);
    } else if (!wide_only && dtype == SPV_BF16 && compact_ok(d) && (size_t)d * 2 <= (size_t)LDS_LIMIT && d <= 8 * PT * C_IT) {
        const Compact c = compact_of(idx, (int64_t)heads * d);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_c16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        hipLaunchKernelGGL(gather_bwd_c16_kernel, dim3(batch), dim3(PT), (size_t)d * 2, st, (const bf16_t*)dg, c.inv16, c.inv_sg, (bf16_t*)dx, heads, d);
    } else if (aligned && (size_t)d * es <= (size_t)LDS_LIMIT && d <= 4 * PT * MAX_IT) {
        const size_t lds = (size_t)d * es;
        if (dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL((gather_bwd_lds_kernel<bf16_t>), dim3(batch), dim3(PT), lds, st, (const bf16_t*)dg, inv, (bf16_t*)dx, heads, d);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
            hipLaunchKernelGGL((gather_bwd_lds_kernel<float>), dim3(batch), dim3(PT), lds, st, (const float*)dg, inv, (float*)dx, heads, d);
        }
    } else if (!SPV_LAB_SET("SPV_GATHER_BWD_PARTS") && dtype == SPV_BF16 && long_row(d) && d % 8 == 0 && (size_t)scat_len(d) * 4 <= (size_t)LDS_LIMIT) {
        // the scatter into output parts (lists built by spv_permut_pack behind the other tables)
        const ScatTail t = scat_of(const_cast<uint32_t*>(idx) + (spv_permut_table_words(heads, d) - scat_words(heads, d)), heads, d);
        const int np = scat_parts(d), L = scat_len(d);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        hipLaunchKernelGGL(gather_bwd_scatter_kernel, dim3(np, batch), dim3(1024), (size_t)L * 4, st, (const bf16_t*)dg, t.off, t.ej, t.et, (bf16_t*)dx, heads, d, L,
                           np);
    } else if (!SPV_LAB_SET("SPV_GATHER_BWD_GLOBAL") && dtype == SPV_BF16 && d % 8 == 0 && (((uintptr_t)dg | (uintptr_t)dx | (uintptr_t)inv) & 15) == 0) {
        // each of a sample's `heads` gradient rows serves only d lookups (the forward's ONE row serves heads x d), so a workgroup stages
        // heads x parts = 24 parts of 151 KB for its 32 768 results: 581 us per Base layer where the global path measures 438 in the
        // kernel statistics -- and still the better step (same job, alternating: 27.73 / 27.88 against 28.17 / 28.28 ms), because it
        // leaves the L2 to the weight-gradient GEMM it runs beside
        const int part = parts_len(d);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_bwd_parts_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
        const dim3 grid((unsigned)((d + PARTS_CH - 1) / PARTS_CH), batch);
        hipLaunchKernelGGL(gather_bwd_parts_kernel, grid, dim3(PARTS_T), (size_t)part * 2, st, (const bf16_t*)dg, inv, (bf16_t*)dx, heads, d, part);
    } else {
        dim3 grid((unsigned)std::min((d + 255) / 256, 1024), batch);
        if (dtype == SPV_BF16)
            hipLaunchKernelGGL((gather_bwd_global_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dg, inv, (bf16_t*)dx, heads, d);
        else
            hipLaunchKernelGGL((gather_bwd_global_kernel<float>), grid, dim3(256), 0, st, (const float*)dg, inv, (float*)dx, heads, d);
    }
    SPV_LAUNCH_CHECK("spv_permut_gather_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Token row 0 only (the last layer of a stack whose consumer reads the CLS row: hip_ops.LAST_LAYER_CLS_ONLY).  Row t of the gathered
// matrix is the chunk [t n, (t + 1) n) of the flattened (heads, d) gather (n = heads * embed: the raw view of reference
// layers.py:72), so row 0 is the first n entries of the forward table: g0[b][c] = +-x[b][idx[c]].  The same launch copies the
// sample's own row 0 (the residual's CLS row).  Backward: dx[b][:] = dx0 in row 0, zero elsewhere, then += the n scattered values
// (a permutation's entries are distinct: no atomics).
namespace {
template <typename T>
__global__ __launch_bounds__(256) void permut_row0_fwd_kernel(const T* __restrict__ x, const uint32_t* __restrict__ idx, T* __restrict__ g0,
                                                              T* __restrict__ x0, int d, int n, int E) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const uint32_t w = idx[c];
    const float v = io<T>::ld(x + (size_t)b * d + (w & 0x7fffffffu));
    io<T>::st(g0 + (size_t)b * n + c, (w >> 31) ? -v : v);
    if (c < E) x0[(size_t)b * E + c] = x[(size_t)b * d + c];
}
// the forward with the sample's row in LDS: one coalesced read of the 65 KB row, the n random two-byte reads served by LDS (from global
// memory they were 34.6 us for 8 MB of output)
template <typename T>
__global__ __launch_bounds__(1024) void permut_row0_fwd_lds_kernel(const T* __restrict__ x, const uint32_t* __restrict__ idx, T* __restrict__ g0,
                                                                   T* __restrict__ x0, int d, int n, int E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char r0f_smem[];
    T* row = reinterpret_cast<T*>(r0f_smem);
    const int b = blockIdx.x;
    constexpr int V = 16 / sizeof(T);
    for (int p = threadIdx.x * V; p < d; p += 1024 * V) {
        const uint4 v = *reinterpret_cast<const uint4*>(x + (size_t)b * d + p);
        *reinterpret_cast<uint4*>(row + p) = v;
        if (p < E) *reinterpret_cast<uint4*>(x0 + (size_t)b * E + p) = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 1024) {
        const uint32_t w = idx[c];
        const float v = io<T>::ld(row + (w & 0x7fffffffu));
        io<T>::st(g0 + (size_t)b * n + c, (w >> 31) ? -v : v);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void permut_row0_init_kernel(const T* __restrict__ dx0, T* __restrict__ dx, int d, int E) {
    // 8 elements (16 or 32 bytes) per thread; d and E are multiples of 8
    const int b = blockIdx.y, p = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (p >= d) return;
    T v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p < E ? dx0[(size_t)b * E + p + u] : T(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) dx[(size_t)b * d + p + u] = v[u];
}
template <typename T>
__global__ __launch_bounds__(256) void permut_row0_scatter_kernel(const T* __restrict__ dg0, const uint32_t* __restrict__ idx, T* __restrict__ dx,
                                                                  int d, int n) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const uint32_t w = idx[c];
    const float v = io<T>::ld(dg0 + (size_t)b * n + c);
    T* dst = dx + (size_t)b * d + (w & 0x7fffffffu);
    io<T>::st(dst, io<T>::ld(dst) + ((w >> 31) ? -v : v));
}
// The same two steps with the sample's whole gradient row in LDS (d elements: 65 KB in bf16 at the Small width): initialise it there,
// scatter-add the n values into it (distinct targets: no atomics), write it out once in 16-byte pieces.  In global memory the scatter
// was 4.2 M two-byte read-modify-writes at random addresses: 272 MB fetched + 111 MB written for a 34 MB result, 91 us (PMC, round 3).
template <typename T>
__global__ __launch_bounds__(1024) void permut_row0_bwd_lds_kernel(const T* __restrict__ dg0, const T* __restrict__ dx0, const uint32_t* __restrict__ idx,
                                                                   T* __restrict__ dx, int d, int n, int E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char r0_smem[];
    T* row = reinterpret_cast<T*>(r0_smem);
    const int b = blockIdx.x;
    constexpr int V = 16 / sizeof(T);   // elements per 16-byte piece; d and E are multiples of 8
    for (int p = threadIdx.x * V; p < d; p += 1024 * V) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p < E) v = *reinterpret_cast<const uint4*>(dx0 + (size_t)b * E + p);
        *reinterpret_cast<uint4*>(row + p) = v;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 1024) {
        const uint32_t w = idx[c];
        const float v = io<T>::ld(dg0 + (size_t)b * n + c);
        T* dst = row + (w & 0x7fffffffu);
        io<T>::st(dst, io<T>::ld(dst) + ((w >> 31) ? -v : v));
    }
    __syncthreads();
    for (int p = threadIdx.x * V; p < d; p += 1024 * V) *reinterpret_cast<uint4*>(dx + (size_t)b * d + p) = *reinterpret_cast<const uint4*>(row + p);
}
}  // namespace

extern "C" int spv_permut_row0_fwd(const void* x, const uint32_t* idx, void* g0, void* x0, int batch, int d, int n, int embed, int dtype,
                                   void* stream) {
    SPV_CHECK(batch > 0 && d > 0 && n > 0 && n <= d && embed > 0 && embed <= d, "spv_permut_row0_fwd: batch=%d d=%d n=%d embed=%d", batch, d, n, embed);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_permut_row0_fwd: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(cdiv(n, 256), batch);
    const size_t row_bytes = (size_t)d * (dtype == SPV_BF16 ? 2 : 4);
    const int vec = dtype == SPV_BF16 ? 8 : 4;
    if (row_bytes <= 150 * 1024 && d % vec == 0 && embed % vec == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)x0 & 15) == 0) {
        if (dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&permut_row0_fwd_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
            hipLaunchKernelGGL(permut_row0_fwd_lds_kernel<bf16_t>, dim3(batch), dim3(1024), row_bytes, st, (const bf16_t*)x, idx, (bf16_t*)g0, (bf16_t*)x0, d, n, embed);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&permut_row0_fwd_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
            hipLaunchKernelGGL(permut_row0_fwd_lds_kernel<float>, dim3(batch), dim3(1024), row_bytes, st, (const float*)x, idx, (float*)g0, (float*)x0, d, n, embed);
        }
    } else if (dtype == SPV_BF16)
        hipLaunchKernelGGL(permut_row0_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, idx, (bf16_t*)g0, (bf16_t*)x0, d, n, embed);
    else
        hipLaunchKernelGGL(permut_row0_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, idx, (float*)g0, (float*)x0, d, n, embed);
    SPV_COUNT_PATH(SPV_PATH_PERMUT_ROW0);
    SPV_LAUNCH_CHECK("spv_permut_row0_fwd");
    return 0;
}

extern "C" int spv_permut_row0_bwd(const void* dg0, const void* dx0, const uint32_t* idx, void* dx, int batch, int d, int n, int embed,
                                   int dtype, void* stream) {
    SPV_CHECK(batch > 0 && d > 0 && n > 0 && n <= d && embed > 0 && embed <= d && d % 8 == 0 && embed % 8 == 0,
              "spv_permut_row0_bwd: batch=%d d=%d n=%d embed=%d (d, embed multiples of 8)", batch, d, n, embed);
    SPV_CHECK(dtype == SPV_F32 || dtype == SPV_BF16, "spv_permut_row0_bwd: bad dtype %d", dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t row_bytes = (size_t)d * (dtype == SPV_BF16 ? 2 : 4);
    if (row_bytes <= 150 * 1024 && (dtype == SPV_BF16 ? d % 8 == 0 : d % 4 == 0) && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dx0 & 15) == 0) {
        // one workgroup per sample, the row in LDS
        if (dtype == SPV_BF16) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&permut_row0_bwd_lds_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
            hipLaunchKernelGGL(permut_row0_bwd_lds_kernel<bf16_t>, dim3(batch), dim3(1024), row_bytes, st, (const bf16_t*)dg0, (const bf16_t*)dx0, idx, (bf16_t*)dx, d,
                               n, embed);
        } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&permut_row0_bwd_lds_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_bytes);
            hipLaunchKernelGGL(permut_row0_bwd_lds_kernel<float>, dim3(batch), dim3(1024), row_bytes, st, (const float*)dg0, (const float*)dx0, idx, (float*)dx, d, n,
                               embed);
        }
    } else if (dtype == SPV_BF16) {
        hipLaunchKernelGGL(permut_row0_init_kernel<bf16_t>, dim3(cdiv(d / 8, 256), batch), dim3(256), 0, st, (const bf16_t*)dx0, (bf16_t*)dx, d, embed);
        hipLaunchKernelGGL(permut_row0_scatter_kernel<bf16_t>, dim3(cdiv(n, 256), batch), dim3(256), 0, st, (const bf16_t*)dg0, idx, (bf16_t*)dx, d, n);
    } else {
        hipLaunchKernelGGL(permut_row0_init_kernel<float>, dim3(cdiv(d / 8, 256), batch), dim3(256), 0, st, (const float*)dx0, (float*)dx, d, embed);
        hipLaunchKernelGGL(permut_row0_scatter_kernel<float>, dim3(cdiv(n, 256), batch), dim3(256), 0, st, (const float*)dg0, idx, (float*)dx, d, n);
    }
    SPV_COUNT_PATH(SPV_PATH_PERMUT_ROW0);
    SPV_LAUNCH_CHECK("spv_permut_row0_bwd");
    return 0;
}
