// HBM-bound layers of the SSD-VGG step for gfx950: pooling, L2 normalisation, the
// multibox head assembly / softmax / hard-negative-mined loss, momentum update.
// All reductions use a fixed order (no float atomics): results are run-to-run identical.
#include "ops.h"
#include "bf16.h"

namespace ssd {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int grid_for(size_t n, int block = 256, int cap = 256 * 16) {
    size_t g = (n + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// =================================================================================
// max pooling
// =================================================================================
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolDesc d, const T* __restrict__ x, T* __restrict__ y) {
    const int C4 = d.C >> 2;
    const size_t total = (size_t)d.B * d.Ho * d.Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t pix = idx / C4;
        const int ow = (int)(pix % d.Wo);
        pix /= d.Wo;
        const int oh = (int)(pix % d.Ho);
        const int b = (int)(pix / d.Ho);
        const int h0 = oh * d.stride - d.pad_h, w0 = ow * d.stride - d.pad_w;
        const float ninf = -__builtin_inff();
        f32x4 m = {ninf, ninf, ninf, ninf};
        for (int kh = 0; kh < d.k; ++kh) {
            const int h = h0 + kh;
            if ((unsigned)h >= (unsigned)d.Hi) continue;
            for (int kw = 0; kw < d.k; ++kw) {
                const int w = w0 + kw;
                if ((unsigned)w >= (unsigned)d.Wi) continue;
                const f32x4 v = ld4t(x + (((size_t)b * d.Hi + h) * d.Wi + w) * d.C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
            }
        }
        st4t(y + idx * 4, m);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolDesc d, const T* __restrict__ x,
                                                          const T* __restrict__ dy, T* __restrict__ dx,
                                                          int accumulate, int relu_mask) {
    const int C4 = d.C >> 2;
    const size_t total = (size_t)d.B * d.Hi * d.Wi * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t pix = idx / C4;
        const int w = (int)(pix % d.Wi);
        pix /= d.Wi;
        const int h = (int)(pix % d.Hi);
        const int b = (int)(pix / d.Hi);
        const f32x4 self = ld4t(x + idx * 4);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (relu_mask && !(self[0] > 0.f) && !(self[1] > 0.f) && !(self[2] > 0.f) && !(self[3] > 0.f)) {
            st4t(dx + idx * 4, g);                    // every component is masked whatever arrives
            continue;
        }
        int n = h + d.pad_h - d.k + 1;
        const int oh_lo = n <= 0 ? 0 : (n + d.stride - 1) / d.stride;
        const int oh_hi = min(d.Ho - 1, (h + d.pad_h) / d.stride);
        n = w + d.pad_w - d.k + 1;
        const int ow_lo = n <= 0 ? 0 : (n + d.stride - 1) / d.stride;
        const int ow_hi = min(d.Wo - 1, (w + d.pad_w) / d.stride);
        for (int oh = oh_lo; oh <= oh_hi; ++oh) {
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int h0 = oh * d.stride - d.pad_h, w0 = ow * d.stride - d.pad_w;
                bool first[4] = {true, true, true, true};
                for (int kh = 0; kh < d.k; ++kh) {
                    const int hh = h0 + kh;
                    if ((unsigned)hh >= (unsigned)d.Hi) continue;
                    for (int kw = 0; kw < d.k; ++kw) {
                        const int ww = w0 + kw;
                        if ((unsigned)ww >= (unsigned)d.Wi) continue;
                        if (hh == h && ww == w) continue;
                        const bool before = hh < h || (hh == h && ww < w);
                        const f32x4 v = ld4t(x + (((size_t)b * d.Hi + hh) * d.Wi + ww) * d.C + c4 * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (before ? v[e] >= self[e] : v[e] > self[e]) first[e] = false;
                    }
                }
                const f32x4 gy = ld4t(dy + (((size_t)b * d.Ho + oh) * d.Wo + ow) * d.C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (first[e]) g[e] += gy[e];
            }
        }
        if (accumulate) g += ld4t(dx + idx * 4);
        if (relu_mask) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = self[e] > 0.f ? g[e] : 0.f;
        }
        st4t(dx + idx * 4, g);
    }
}

// ---- 2x2 stride-2 SAME pooling: never overlaps and never pads before the image -----------------------------------------
// One workgroup per output row (b, oh); a thread owns one window x V channels, V = 16 bytes' worth (4 fp32 / 8 bf16).
// All of a window's loads are issued up front, unconditionally, from clamped coordinates (a cell outside the image re-reads
// its in-image neighbour and is ignored by the `ok` test afterwards): a predicated load makes the loads wait for each other
// (s_waitcnt vmcnt(0) after each).  Index math is 32-bit, the row split is per workgroup (scalar).
template <typename T, int V> struct RawV;
template <> struct RawV<float, 4> { typedef f32x4 type; };
template <> struct RawV<bf16_t, 4> { typedef u32x2 type; };
template <> struct RawV<bf16_t, 8> { typedef u32x4 type; };
__device__ __forceinline__ void unpack(f32x4 r, float (&f)[4]) { f[0] = r[0]; f[1] = r[1]; f[2] = r[2]; f[3] = r[3]; }
__device__ __forceinline__ void unpack(u32x2 r, float (&f)[4]) { f[0] = lo2f(r[0]); f[1] = hi2f(r[0]); f[2] = lo2f(r[1]); f[3] = hi2f(r[1]); }
__device__ __forceinline__ void unpack(u32x4 r, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = lo2f(r[i]); f[2 * i + 1] = hi2f(r[i]); }
}
__device__ __forceinline__ void pack(const float (&f)[4], f32x4& r) { r = f32x4{f[0], f[1], f[2], f[3]}; }
__device__ __forceinline__ void pack(const float (&f)[4], u32x2& r) { r = u32x2{pack2(f[0], f[1]), pack2(f[2], f[3])}; }
__device__ __forceinline__ void pack(const float (&f)[8], u32x4& r) {
    r = u32x4{pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
}

struct PoolRow {      // the output row of this workgroup
    int b, oh, h0, h1;
    bool okh;
    __device__ __forceinline__ PoolRow(const PoolDesc& d) {
        oh = blockIdx.x % d.Ho; b = blockIdx.x / d.Ho;
        h0 = oh * 2; okh = h0 + 1 < d.Hi; h1 = okh ? h0 + 1 : h0;
    }
};

template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(PoolDesc d, const T* __restrict__ x,
                                                             const T* __restrict__ dy, T* __restrict__ dx,
                                                             int accumulate, int relu_mask) {
    typedef typename RawV<T, V>::type raw_t;
    const PoolRow R(d);
    const unsigned CV = d.C / V, items = d.Wo * CV;
    const size_t in0 = ((size_t)R.b * d.Hi + R.h0) * d.Wi * d.C, in1 = ((size_t)R.b * d.Hi + R.h1) * d.Wi * d.C;
    const T* dyr = dy + (size_t)blockIdx.x * d.Wo * d.C;
    for (unsigned i = threadIdx.x; i < items; i += 256) {
        const unsigned ow = i / CV, cv = i - ow * CV;
        const int w0 = ow * 2;
        const bool okw = w0 + 1 < d.Wi;
        const int w1 = okw ? w0 + 1 : w0;
        const size_t o[4] = {in0 + (size_t)w0 * d.C + cv * V, in0 + (size_t)w1 * d.C + cv * V,
                             in1 + (size_t)w0 * d.C + cv * V, in1 + (size_t)w1 * d.C + cv * V};
        const bool ok[4] = {true, okw, R.okh, R.okh && okw};
        raw_t rx[4], ro[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rx[q] = *reinterpret_cast<const raw_t*>(x + o[q]);
        const raw_t rg = *reinterpret_cast<const raw_t*>(dyr + (size_t)i * V);
        if (accumulate) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ro[q] = *reinterpret_cast<const raw_t*>(dx + o[q]);
        }
        float v[4][V], gy[V], g[4][V];
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack(rx[q], v[q]);
        unpack(rg, gy);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (accumulate) unpack(ro[q], g[q]);
            else {
#pragma unroll
                for (int e = 0; e < V; ++e) g[q][e] = 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            int a = 0;
            float m = v[0][e];                       // cell 0 is always inside the image
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (ok[q] && v[q][e] > m) { m = v[q][e]; a = q; }   // strict: the first maximum wins
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (a == q) g[q][e] += gy[e];
                if (relu_mask && !(v[q][e] > 0.f)) g[q][e] = 0.f;   // every component is masked whatever arrives
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            raw_t out;
            pack(g[q], out);
            *reinterpret_cast<raw_t*>(dx + o[q]) = out;
        }
    }
}

// The same pooling with a 12-bit record per (window, 4 channels) written by the forward pass: per channel the
// first maximum's cell (2 bits) and whether that maximum is positive (1 bit).  Backward then needs neither the
// input tensor (argmax) nor its sign (relu mask of the producing conv): it reads record + dy and writes dx,
// 0.58x the bytes of the kernel above.  Valid when the pooled tensor has no other consumer (nothing to accumulate).
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2x2_fwd_rec_kernel(PoolDesc d, const T* __restrict__ x, T* __restrict__ y,
                                                                 unsigned short* __restrict__ rec) {
    typedef typename RawV<T, V>::type raw_t;
    const PoolRow R(d);
    const unsigned CV = d.C / V, items = d.Wo * CV;
    const T* x0 = x + ((size_t)R.b * d.Hi + R.h0) * d.Wi * d.C;
    const T* x1 = x + ((size_t)R.b * d.Hi + R.h1) * d.Wi * d.C;
    T* yr = y + (size_t)blockIdx.x * d.Wo * d.C;
    unsigned short* rr = rec + (size_t)blockIdx.x * d.Wo * (d.C / 4);
    for (unsigned i = threadIdx.x; i < items; i += 256) {
        const unsigned ow = i / CV, cv = i - ow * CV;
        const int w0 = ow * 2;
        const bool okw = w0 + 1 < d.Wi;
        const int w1 = okw ? w0 + 1 : w0;
        const bool ok[4] = {true, okw, R.okh, R.okh && okw};
        raw_t rx[4];
        rx[0] = *reinterpret_cast<const raw_t*>(x0 + (size_t)w0 * d.C + cv * V);
        rx[1] = *reinterpret_cast<const raw_t*>(x0 + (size_t)w1 * d.C + cv * V);
        rx[2] = *reinterpret_cast<const raw_t*>(x1 + (size_t)w0 * d.C + cv * V);
        rx[3] = *reinterpret_cast<const raw_t*>(x1 + (size_t)w1 * d.C + cv * V);
        float v[4][V], m[V];
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack(rx[q], v[q]);
        unsigned r[V / 4];
#pragma unroll
        for (int gq = 0; gq < V / 4; ++gq) r[gq] = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            unsigned a = 0;
            float mm = v[0][e];                      // cell 0 is always inside the image
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (ok[q] && v[q][e] > mm) { mm = v[q][e]; a = q; }   // strict: the first maximum wins
            m[e] = mm;
            r[e / 4] |= (a | (mm > 0.f ? 4u : 0u)) << (3 * (e & 3));
        }
        raw_t out;
        pack(m, out);
        *reinterpret_cast<raw_t*>(yr + (size_t)i * V) = out;
        if constexpr (V == 8) *reinterpret_cast<unsigned*>(rr + (size_t)i * 2) = r[0] | (r[1] << 16);
        else rr[i] = (unsigned short)r[0];
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool2x2_bwd_rec_kernel(PoolDesc d, const unsigned short* __restrict__ rec,
                                                                 const T* __restrict__ dy, T* __restrict__ dx, int relu_mask) {
    typedef typename RawV<T, V>::type raw_t;
    const PoolRow R(d);
    const unsigned CV = d.C / V, items = d.Wo * CV;
    T* d0 = dx + ((size_t)R.b * d.Hi + R.h0) * d.Wi * d.C;
    T* d1 = dx + ((size_t)R.b * d.Hi + R.h1) * d.Wi * d.C;
    const T* dyr = dy + (size_t)blockIdx.x * d.Wo * d.C;
    const unsigned short* rr = rec + (size_t)blockIdx.x * d.Wo * (d.C / 4);
    for (unsigned i = threadIdx.x; i < items; i += 256) {
        const unsigned ow = i / CV, cv = i - ow * CV;
        const int w0 = ow * 2;
        const bool okw = w0 + 1 < d.Wi;
        unsigned rw;
        if constexpr (V == 8) rw = *reinterpret_cast<const unsigned*>(rr + (size_t)i * 2);
        else rw = rr[i];
        const raw_t rg = *reinterpret_cast<const raw_t*>(dyr + (size_t)i * V);
        float gy[V];
        unpack(rg, gy);
        const bool ok[4] = {true, okw, R.okh, R.okh && okw};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            float g[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const unsigned re = (rw >> (16 * (e / 4) + 3 * (e & 3))) & 7u;
                g[e] = ((re & 3u) == (unsigned)q && (!relu_mask || (re & 4u))) ? gy[e] : 0.f;
            }
            raw_t out;
            pack(g, out);
            *reinterpret_cast<raw_t*>((q >> 1 ? d1 : d0) + (size_t)(w0 + (q & 1)) * d.C + cv * V) = out;
        }
    }
}

// Overlapping pooling (3x3 stride 1, mod_pool5): pass A finds every window's first maximum once
// (its scan-order cell index, one byte per channel); pass B lets every input cell collect dy from
// the <= 9 windows whose recorded maximum it is.  27 loads per cell instead of 81.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_argmax_kernel(PoolDesc d, const T* __restrict__ x, unsigned* __restrict__ arg) {
    const int C4 = d.C >> 2;
    const size_t total = (size_t)d.B * d.Ho * d.Wo * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t pix = idx / C4;
        const int ow = (int)(pix % d.Wo);
        pix /= d.Wo;
        const int oh = (int)(pix % d.Ho);
        const int b = (int)(pix / d.Ho);
        const int h0 = oh * d.stride - d.pad_h, w0 = ow * d.stride - d.pad_w;
        const float ninf = -__builtin_inff();
        f32x4 m = {ninf, ninf, ninf, ninf};
        unsigned a[4] = {255u, 255u, 255u, 255u};
        for (int kh = 0; kh < d.k; ++kh) {
            const int h = h0 + kh;
            if ((unsigned)h >= (unsigned)d.Hi) continue;
            for (int kw = 0; kw < d.k; ++kw) {
                const int w = w0 + kw;
                if ((unsigned)w >= (unsigned)d.Wi) continue;
                const f32x4 v = ld4t(x + (((size_t)b * d.Hi + h) * d.Wi + w) * d.C + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v[e] > m[e] || a[e] == 255u) { m[e] = v[e]; a[e] = (unsigned)(kh * d.k + kw); }   // strict: first maximum
            }
        }
        arg[idx] = a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24);
    }
}

// K x K windows (K = 2, 3) with every tap's load issued before the first comparison: coordinates outside the image are
// clamped for the address and excluded by `ok` afterwards.  Writes the maxima (y != nullptr) and / or the first
// maximum's scan-order tap per channel (arg != nullptr: the record pass A of the backward below).  32-bit index math.
template <typename T, int K>
__global__ __launch_bounds__(256) void maxpool_taps_kernel(PoolDesc d, const T* __restrict__ x, T* __restrict__ y,
                                                           unsigned* __restrict__ arg) {
    typedef typename RawV<T, 4>::type raw_t;
    const unsigned C4 = d.C >> 2, total = (unsigned)d.B * d.Ho * d.Wo * C4;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        unsigned pix = idx / C4;
        const unsigned c4 = idx - pix * C4;
        const int ow = (int)(pix % d.Wo);
        pix /= d.Wo;
        const int oh = (int)(pix % d.Ho);
        const int b = (int)(pix / d.Ho);
        const int h0 = oh * d.stride - d.pad_h, w0 = ow * d.stride - d.pad_w;
        raw_t r[K * K];
        bool ok[K * K];
#pragma unroll
        for (int kh = 0; kh < K; ++kh)
#pragma unroll
            for (int kw = 0; kw < K; ++kw) {
                const int h = h0 + kh, w = w0 + kw;
                ok[kh * K + kw] = (unsigned)h < (unsigned)d.Hi && (unsigned)w < (unsigned)d.Wi;
                const int hc = min(max(h, 0), d.Hi - 1), wc = min(max(w, 0), d.Wi - 1);
                r[kh * K + kw] = *reinterpret_cast<const raw_t*>(x + (((size_t)b * d.Hi + hc) * d.Wi + wc) * d.C + c4 * 4);
            }
        const float ninf = -__builtin_inff();
        float m[4] = {ninf, ninf, ninf, ninf};
        unsigned a[4] = {255u, 255u, 255u, 255u};
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
            float v[4];
            unpack(r[t], v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ok[t] && (v[e] > m[e] || a[e] == 255u)) { m[e] = v[e]; a[e] = (unsigned)t; }   // strict: first maximum
        }
        if (y) {
            raw_t out;
            pack(m, out);
            *reinterpret_cast<raw_t*>(y + (size_t)idx * 4) = out;
        }
        if (arg) arg[idx] = a[0] | (a[1] << 8) | (a[2] << 16) | (a[3] << 24);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_arg_kernel(PoolDesc d, const T* __restrict__ x, const unsigned* __restrict__ arg,
                                                              const T* __restrict__ dy, T* __restrict__ dx,
                                                              int accumulate, int relu_mask) {
    const int C4 = d.C >> 2;
    const size_t total = (size_t)d.B * d.Hi * d.Wi * C4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % C4);
        size_t pix = idx / C4;
        const int w = (int)(pix % d.Wi);
        pix /= d.Wi;
        const int h = (int)(pix % d.Hi);
        const int b = (int)(pix / d.Hi);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        int n = h + d.pad_h - d.k + 1;
        const int oh_lo = n <= 0 ? 0 : (n + d.stride - 1) / d.stride;
        const int oh_hi = min(d.Ho - 1, (h + d.pad_h) / d.stride);
        n = w + d.pad_w - d.k + 1;
        const int ow_lo = n <= 0 ? 0 : (n + d.stride - 1) / d.stride;
        const int ow_hi = min(d.Wo - 1, (w + d.pad_w) / d.stride);
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const size_t o = (((size_t)b * d.Ho + oh) * d.Wo + ow) * C4 + c4;
                const unsigned me = (unsigned)((h - (oh * d.stride - d.pad_h)) * d.k + (w - (ow * d.stride - d.pad_w)));
                const unsigned a = arg[o];
                const f32x4 gy = ld4t(dy + o * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (((a >> (8 * e)) & 255u) == me) g[e] += gy[e];
            }
        if (accumulate) g += ld4t(dx + idx * 4);
        if (relu_mask) {
            const f32x4 self = ld4t(x + idx * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = self[e] > 0.f ? g[e] : 0.f;
        }
        st4t(dx + idx * 4, g);
    }
}

// Pass B for stride 1 (mod_pool5): input cell (h, w) is tap (j, i) of window (h + pad - j, w + pad - i).  The K x K records
// and gradients are loaded up front from clamped window coordinates, then summed in the order of the kernel above
// (windows ascending = taps descending).
template <typename T, int K>
__global__ __launch_bounds__(256) void maxpool_bwd_arg_s1_kernel(PoolDesc d, const T* __restrict__ x, const unsigned* __restrict__ arg,
                                                                 const T* __restrict__ dy, T* __restrict__ dx,
                                                                 int accumulate, int relu_mask) {
    typedef typename RawV<T, 4>::type raw_t;
    const unsigned C4 = d.C >> 2, total = (unsigned)d.B * d.Hi * d.Wi * C4;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        unsigned pix = idx / C4;
        const unsigned c4 = idx - pix * C4;
        const int w = (int)(pix % d.Wi);
        pix /= d.Wi;
        const int h = (int)(pix % d.Hi);
        const int b = (int)(pix / d.Hi);
        raw_t rg[K * K], rself, rold;
        unsigned ra[K * K];
        bool ok[K * K];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int oh = h + d.pad_h - j, ow = w + d.pad_w - i;
                ok[j * K + i] = (unsigned)oh < (unsigned)d.Ho && (unsigned)ow < (unsigned)d.Wo;
                const int ohc = min(max(oh, 0), d.Ho - 1), owc = min(max(ow, 0), d.Wo - 1);
                const size_t o = (((size_t)b * d.Ho + ohc) * d.Wo + owc) * C4 + c4;
                ra[j * K + i] = arg[o];
                rg[j * K + i] = *reinterpret_cast<const raw_t*>(dy + o * 4);
            }
        if (relu_mask) rself = *reinterpret_cast<const raw_t*>(x + (size_t)idx * 4);
        if (accumulate) rold = *reinterpret_cast<const raw_t*>(dx + (size_t)idx * 4);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = K * K - 1; t >= 0; --t) {
            float gy[4];
            unpack(rg[t], gy);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ok[t] && ((ra[t] >> (8 * e)) & 255u) == (unsigned)t) g[e] += gy[e];
        }
        if (accumulate) {
            float old[4];
            unpack(rold, old);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] += old[e];
        }
        if (relu_mask) {
            float self[4];
            unpack(rself, self);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = self[e] > 0.f ? g[e] : 0.f;
        }
        raw_t out;
        pack(g, out);
        *reinterpret_cast<raw_t*>(dx + (size_t)idx * 4) = out;
    }
}

template <typename T>
static void maxpool_fwd_t(const PoolDesc& d, const T* x, T* y, hipStream_t s) {
    SSD_REQUIRE(d.C % 4 == 0, "maxpool: C must be a multiple of 4");
    const size_t total = (size_t)d.B * d.Ho * d.Wo * (d.C / 4);
    ProfScope prof("maxpool_fwd", 0.0, sizeof(T) * (double)d.C * d.B * ((double)d.Hi * d.Wi + (double)d.Ho * d.Wo), s);
    const bool small_idx = (double)d.B * d.Hi * d.Wi * d.C < 2.0e9;
    if (d.k == 3 && small_idx)
        hipLaunchKernelGGL((maxpool_taps_kernel<T, 3>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x, y, (unsigned*)nullptr);
    else if (d.k == 2 && small_idx)
        hipLaunchKernelGGL((maxpool_taps_kernel<T, 2>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x, y, (unsigned*)nullptr);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x, y);
    HIP_OK(hipGetLastError());
}

bool maxpool_rec_applicable(const PoolDesc& d) { return d.k == 2 && d.stride == 2 && d.pad_h == 0 && d.pad_w == 0 && d.C % 4 == 0; }
size_t maxpool_rec_bytes(const PoolDesc& d) { return (size_t)d.B * d.Ho * d.Wo * (d.C / 4) * sizeof(unsigned short); }

template <typename T>
static void maxpool_fwd_rec_t(const PoolDesc& d, const T* x, T* y, void* rec, hipStream_t s) {
    SSD_REQUIRE(maxpool_rec_applicable(d), "maxpool record: 2x2 stride-2 pooling without leading padding only");
    const size_t total = (size_t)d.B * d.Ho * d.Wo * (d.C / 4);
    ProfScope prof("maxpool_fwd", 0.0, sizeof(T) * (double)d.C * d.B * ((double)d.Hi * d.Wi + (double)d.Ho * d.Wo) + 2.0 * total, s);
    if (sizeof(T) == 2 && d.C % 8 == 0)
        hipLaunchKernelGGL((maxpool2x2_fwd_rec_kernel<T, sizeof(T) == 2 ? 8 : 4>), dim3(d.B * d.Ho), dim3(256), 0, s, d, x, y, (unsigned short*)rec);
    else
        hipLaunchKernelGGL((maxpool2x2_fwd_rec_kernel<T, 4>), dim3(d.B * d.Ho), dim3(256), 0, s, d, x, y, (unsigned short*)rec);
    HIP_OK(hipGetLastError());
}
void maxpool_fwd_rec(const PoolDesc& d, const float* x, float* y, void* rec, hipStream_t s) { maxpool_fwd_rec_t(d, x, y, rec, s); }
void maxpool_fwd_rec(const PoolDesc& d, const bf16_t* x, bf16_t* y, void* rec, hipStream_t s) { maxpool_fwd_rec_t(d, x, y, rec, s); }

template <typename T>
static void maxpool_bwd_rec_t(const PoolDesc& d, const void* rec, const T* dy, T* dx, bool relu_mask, hipStream_t s) {
    const size_t total = (size_t)d.B * d.Ho * d.Wo * (d.C / 4);
    ProfScope prof("maxpool_bwd", 0.0, sizeof(T) * (double)d.C * d.B * ((double)d.Hi * d.Wi + (double)d.Ho * d.Wo) + 2.0 * total, s);
    if (sizeof(T) == 2 && d.C % 8 == 0)
        hipLaunchKernelGGL((maxpool2x2_bwd_rec_kernel<T, sizeof(T) == 2 ? 8 : 4>), dim3(d.B * d.Ho), dim3(256), 0, s, d, (const unsigned short*)rec, dy,
                           dx, (int)relu_mask);
    else
        hipLaunchKernelGGL((maxpool2x2_bwd_rec_kernel<T, 4>), dim3(d.B * d.Ho), dim3(256), 0, s, d, (const unsigned short*)rec, dy, dx, (int)relu_mask);
    HIP_OK(hipGetLastError());
}
void maxpool_bwd_rec(const PoolDesc& d, const void* rec, const float* dy, float* dx, bool relu_mask, hipStream_t s) {
    maxpool_bwd_rec_t(d, rec, dy, dx, relu_mask, s);
}
void maxpool_bwd_rec(const PoolDesc& d, const void* rec, const bf16_t* dy, bf16_t* dx, bool relu_mask, hipStream_t s) {
    maxpool_bwd_rec_t(d, rec, dy, dx, relu_mask, s);
}

size_t maxpool_bwd_ws_bytes(const PoolDesc& d) {
    return (d.k == 2 && d.stride == 2) ? 0 : (size_t)d.B * d.Ho * d.Wo * (d.C / 4) * sizeof(unsigned);
}

void maxpool_fwd(const PoolDesc& d, const float* x, float* y, hipStream_t s) { maxpool_fwd_t(d, x, y, s); }
void maxpool_fwd(const PoolDesc& d, const bf16_t* x, bf16_t* y, hipStream_t s) { maxpool_fwd_t(d, x, y, s); }

// 3x3 stride-1 pooling (mod_pool5) whose forward pass keeps every window's first-maximum tap (the scratch of maxpool_bwd, 4
// bytes per window and 4 channels): backward then runs its second pass only -- the first one, which recomputes exactly this,
// sat on the data-gradient chain's critical path (28 of 51 us at batch 32, profiles/r05_l_timeline_merged_tail_bf16.txt)
bool maxpool_arg_applicable(const PoolDesc& d) { return d.k == 3 && d.stride == 1 && d.C % 4 == 0 && (double)d.B * d.Hi * d.Wi * d.C < 2.0e9; }
template <typename T>
static void maxpool_fwd_arg_t(const PoolDesc& d, const T* x, T* y, void* arg, hipStream_t s) {
    SSD_REQUIRE(maxpool_arg_applicable(d) && arg != nullptr, "maxpool_fwd_arg: 3x3 stride-1 pooling with a record buffer");
    const size_t total = (size_t)d.B * d.Ho * d.Wo * (d.C / 4);
    ProfScope prof("maxpool_fwd", 0.0, sizeof(T) * (double)d.C * d.B * ((double)d.Hi * d.Wi + (double)d.Ho * d.Wo) + 4.0 * total, s);
    hipLaunchKernelGGL((maxpool_taps_kernel<T, 3>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x, y, (unsigned*)arg);
    HIP_OK(hipGetLastError());
}
template <typename T>
static void maxpool_bwd_arg_t(const PoolDesc& d, const T* x, const void* arg, const T* dy, T* dx, bool accumulate, bool relu_mask, hipStream_t s) {
    SSD_REQUIRE(maxpool_arg_applicable(d) && arg != nullptr, "maxpool_bwd_arg: 3x3 stride-1 pooling with the forward pass' record");
    const size_t total = (size_t)d.B * d.Hi * d.Wi * (d.C / 4);
    ProfScope prof("maxpool_bwd", 0.0, sizeof(T) * (double)d.C * d.B * (2.0 * d.Hi * d.Wi + (double)d.Ho * d.Wo), s);
    hipLaunchKernelGGL((maxpool_bwd_arg_s1_kernel<T, 3>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x, (const unsigned*)arg, dy, dx,
                       (int)accumulate, (int)relu_mask);
    HIP_OK(hipGetLastError());
}
void maxpool_fwd_arg(const PoolDesc& d, const float* x, float* y, void* arg, hipStream_t s) { maxpool_fwd_arg_t(d, x, y, arg, s); }
void maxpool_fwd_arg(const PoolDesc& d, const bf16_t* x, bf16_t* y, void* arg, hipStream_t s) { maxpool_fwd_arg_t(d, x, y, arg, s); }
void maxpool_bwd_arg(const PoolDesc& d, const float* x, const void* arg, const float* dy, float* dx, bool accumulate, bool relu_mask, hipStream_t s) {
    maxpool_bwd_arg_t(d, x, arg, dy, dx, accumulate, relu_mask, s);
}
void maxpool_bwd_arg(const PoolDesc& d, const bf16_t* x, const void* arg, const bf16_t* dy, bf16_t* dx, bool accumulate, bool relu_mask, hipStream_t s) {
    maxpool_bwd_arg_t(d, x, arg, dy, dx, accumulate, relu_mask, s);
}

template <typename T>
static void maxpool_bwd_t(const PoolDesc& d, const T* x, const T* dy, T* dx, bool accumulate, bool relu_mask, void* ws,
                          hipStream_t s) {
    SSD_REQUIRE(d.C % 4 == 0, "maxpool: C must be a multiple of 4");
    const size_t total = (size_t)d.B * d.Hi * d.Wi * (d.C / 4);
    ProfScope prof("maxpool_bwd", 0.0, sizeof(T) * (double)d.C * d.B * (2.0 * d.Hi * d.Wi + (double)d.Ho * d.Wo), s);
    if (d.k == 2 && d.stride == 2 && d.pad_h == 0 && d.pad_w == 0) {
        if (sizeof(T) == 2 && d.C % 8 == 0)
            hipLaunchKernelGGL((maxpool2x2_bwd_kernel<T, sizeof(T) == 2 ? 8 : 4>), dim3(d.B * d.Ho), dim3(256), 0, s, d, x, dy, dx, (int)accumulate,
                               (int)relu_mask);
        else
            hipLaunchKernelGGL((maxpool2x2_bwd_kernel<T, 4>), dim3(d.B * d.Ho), dim3(256), 0, s, d, x, dy, dx, (int)accumulate, (int)relu_mask);
        HIP_OK(hipGetLastError());
        return;
    }
    if (ws) {
        const size_t nwin = (size_t)d.B * d.Ho * d.Wo * (d.C / 4);
        if (d.k == 3 && d.stride == 1 && (double)d.B * d.Hi * d.Wi * d.C < 2.0e9) {      // mod_pool5: every load up front
            hipLaunchKernelGGL((maxpool_taps_kernel<T, 3>), dim3(grid_for(nwin, 256, 256 * 32)), dim3(256), 0, s, d, x, (T*)nullptr, (unsigned*)ws);
            hipLaunchKernelGGL((maxpool_bwd_arg_s1_kernel<T, 3>), dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x,
                               (const unsigned*)ws, dy, dx, (int)accumulate, (int)relu_mask);
        } else {
            hipLaunchKernelGGL(maxpool_argmax_kernel<T>, dim3(grid_for(nwin, 256, 256 * 32)), dim3(256), 0, s, d, x, (unsigned*)ws);
            hipLaunchKernelGGL(maxpool_bwd_arg_kernel<T>, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x,
                               (const unsigned*)ws, dy, dx, (int)accumulate, (int)relu_mask);
        }
        HIP_OK(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, d, x, dy, dx,
                       (int)accumulate, (int)relu_mask);
    HIP_OK(hipGetLastError());
}

void maxpool_bwd(const PoolDesc& d, const float* x, const float* dy, float* dx, bool accumulate, bool relu_mask, void* ws,
                 hipStream_t s) {
    maxpool_bwd_t(d, x, dy, dx, accumulate, relu_mask, ws, s);
}
void maxpool_bwd(const PoolDesc& d, const bf16_t* x, const bf16_t* dy, bf16_t* dx, bool accumulate, bool relu_mask, void* ws,
                 hipStream_t s) {
    maxpool_bwd_t(d, x, dy, dx, accumulate, relu_mask, ws, s);
}

// =================================================================================
// L2 normalisation over channels: one wave64 per pixel, each lane owns float4 chunks
// =================================================================================
constexpr int L2_MAXJ = 4;   // C <= 1024

// Four channels as they lie in memory (fp32: 16 bytes, bf16: 8 bytes).  The kernels below load RAW values for all their
// pixels first (unconditionally, from clamped addresses) and convert / mask them in a second pass: a conversion or a select
// on a loaded value inside the load group makes every load wait for the one before it (s_waitcnt vmcnt(0) each; the first
// version of l2norm_bwd ran its 8 loads per iteration as 8 dependent round trips: 80 us for 142 MB).
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef f32x4 type; };
template <> struct Raw4<bf16_t> { typedef u32x2 type; };
template <typename T>
__device__ __forceinline__ typename Raw4<T>::type ld4raw(const T* p) { return *reinterpret_cast<const typename Raw4<T>::type*>(p); }
__device__ __forceinline__ f32x4 cvt4(f32x4 r) { return r; }
__device__ __forceinline__ f32x4 cvt4(u32x2 w) { return f32x4{lo2f(w[0]), hi2f(w[0]), lo2f(w[1]), hi2f(w[1])}; }

template <typename T, int J>      // J = ceil(C / 256) chunks of 4 channels per lane
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(int npix, int C, const T* __restrict__ x,
                                                         const float* __restrict__ scale, T* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int PX = 2;
    int cc[J];
    f32x4 sc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = lane * 4 + 256 * j;
        cc[j] = c < C ? c : 0;
        sc[j] = ld4(scale + cc[j]);
    }
    for (int pix0 = wave * PX; pix0 < npix; pix0 += nwaves * PX) {
        typename Raw4<T>::type raw[PX][J];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int pix = min(pix0 + q, npix - 1);
#pragma unroll
            for (int j = 0; j < J; ++j) raw[q][j] = ld4raw(x + (size_t)pix * C + cc[j]);
        }
        f32x4 v[PX][J];
        float ss[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            ss[q] = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                v[q][j] = cvt4(raw[q][j]);
                if (lane * 4 + 256 * j >= C) v[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                ss[q] += v[q][j][0] * v[q][j][0] + v[q][j][1] * v[q][j][1] + v[q][j][2] * v[q][j][2] + v[q][j][3] * v[q][j][3];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int q = 0; q < PX; ++q) ss[q] += __shfl_xor(ss[q], o, 64);
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int pix = pix0 + q;
            if (pix >= npix) continue;
            const float r = rsqrtf(fmaxf(ss[q], 1e-12f));
#pragma unroll
            for (int j = 0; j < J; ++j)
                if (lane * 4 + 256 * j < C) st4t(y + (size_t)pix * C + cc[j], sc[j] * v[q][j] * r);
        }
    }
}

// dx = scale*dy*r - x * (sum_c scale*dy*x) * r^3  (when sum x^2 > eps; else the norm is the
// constant sqrt(eps) and only the first term remains).  dscale partials per block -> ws.
template <typename T, int J>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(int npix, int C, const T* __restrict__ x,
                                                         const float* __restrict__ scale, const T* __restrict__ dy,
                                                         T* __restrict__ dx, float* __restrict__ ws) {
    __shared__ float red[4][256 * J];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int PX = J <= 2 ? 4 : 2;      // pixels per wave and iteration: 2 PX J loads in flight, their shuffle trees overlap
    int cc[J];
    f32x4 ds[J], sc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int c = lane * 4 + 256 * j;
        cc[j] = c < C ? c : 0;
        ds[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        sc[j] = ld4(scale + cc[j]);
        if (c >= C) sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int pix0 = wave * PX; pix0 < npix; pix0 += nwaves * PX) {
        typename Raw4<T>::type rx[PX][J], rg[PX][J];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int pix = min(pix0 + q, npix - 1);
#pragma unroll
            for (int j = 0; j < J; ++j) {
                rx[q][j] = ld4raw(x + (size_t)pix * C + cc[j]);
                rg[q][j] = ld4raw(dy + (size_t)pix * C + cc[j]);
            }
        }
        f32x4 v[PX][J], g[PX][J];
        float ss[PX], t[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            ss[q] = t[q] = 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                v[q][j] = cvt4(rx[q][j]);
                g[q][j] = cvt4(rg[q][j]);
                if (lane * 4 + 256 * j >= C) v[q][j] = g[q][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ss[q] += v[q][j][e] * v[q][j][e];
                    t[q] += sc[j][e] * g[q][j][e] * v[q][j][e];
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                ss[q] += __shfl_xor(ss[q], o, 64);
                t[q] += __shfl_xor(t[q], o, 64);
            }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int pix = pix0 + q;
            if (pix >= npix) continue;
            const float r = rsqrtf(fmaxf(ss[q], 1e-12f));
            const float k = ss[q] > 1e-12f ? t[q] * r * r * r : 0.f;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                if (lane * 4 + 256 * j < C) st4t(dx + (size_t)pix * C + cc[j], sc[j] * g[q][j] * r - v[q][j] * k);
                ds[j] += g[q][j] * v[q][j] * r;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wib][lane * 4 + 256 * j + e] = ds[j][e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        ws[(size_t)blockIdx.x * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}

// column sums of ws[nrows][C] in a fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ ws, int nrows, int C, float* __restrict__ out) {
    // a workgroup = 16 columns x 16 row groups; a thread sums its rows (r = group, group + 16, ...) four independent loads at a
    // time (the first version walked 128 rows per wave with one dependent load each: 25 us for 1 MB), fixed order throughout
    __shared__ float red[16][17];
    const int col = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < C) {
        int r = rg;
        for (; r + 48 < nrows; r += 64) {
            const float a0 = ws[(size_t)r * C + c], a1 = ws[(size_t)(r + 16) * C + c];
            const float a2 = ws[(size_t)(r + 32) * C + c], a3 = ws[(size_t)(r + 48) * C + c];
            s0 += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; r < nrows; r += 16) s0 += ws[(size_t)r * C + c];
    }
    red[rg][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][col];
        out[c] = t;
    }
}

static int l2_blocks(int npix) {      // (2048 workgroups were measured: the pixel loop gains nothing, the column sums lose)
    int b = (npix + 3) / 4;
    return b > 512 ? 512 : (b < 1 ? 1 : b);
}

template <typename T>
static void l2norm_fwd_t(int npix, int C, const T* x, const float* scale, T* y, hipStream_t s) {
    SSD_REQUIRE(C % 4 == 0 && C <= 1024, "l2norm: C must be a multiple of 4 and <= 1024");
    int b = (npix + 7) / 8;      // 4 waves x 2 pixels per workgroup and iteration
    if (b > 4096) b = 4096;
    ProfScope prof("l2norm_fwd", 0.0, 2.0 * sizeof(T) * npix * C, s);
    switch ((C + 255) / 256) {
    case 1: hipLaunchKernelGGL((l2norm_fwd_kernel<T, 1>), dim3(b), dim3(256), 0, s, npix, C, x, scale, y); break;
    case 2: hipLaunchKernelGGL((l2norm_fwd_kernel<T, 2>), dim3(b), dim3(256), 0, s, npix, C, x, scale, y); break;
    default: hipLaunchKernelGGL((l2norm_fwd_kernel<T, L2_MAXJ>), dim3(b), dim3(256), 0, s, npix, C, x, scale, y); break;
    }
    HIP_OK(hipGetLastError());
}
void l2norm_fwd(int npix, int C, const float* x, const float* scale, float* y, hipStream_t s) { l2norm_fwd_t(npix, C, x, scale, y, s); }
void l2norm_fwd(int npix, int C, const bf16_t* x, const float* scale, bf16_t* y, hipStream_t s) { l2norm_fwd_t(npix, C, x, scale, y, s); }

size_t l2norm_bwd_ws_floats(int npix, int C) { return (size_t)l2_blocks(npix) * C; }

template <typename T>
static void l2norm_bwd_t(int npix, int C, const T* x, const float* scale, const T* dy, T* dx, float* dscale, float* ws,
                         hipStream_t s) {
    SSD_REQUIRE(C % 4 == 0 && C <= 1024, "l2norm: C must be a multiple of 4 and <= 1024");
    const int nb = l2_blocks(npix);
    ProfScope prof("l2norm_bwd", 0.0, 3.0 * sizeof(T) * npix * C, s);
    switch ((C + 255) / 256) {
    case 1: hipLaunchKernelGGL((l2norm_bwd_kernel<T, 1>), dim3(nb), dim3(256), 0, s, npix, C, x, scale, dy, dx, ws); break;
    case 2: hipLaunchKernelGGL((l2norm_bwd_kernel<T, 2>), dim3(nb), dim3(256), 0, s, npix, C, x, scale, dy, dx, ws); break;
    default: hipLaunchKernelGGL((l2norm_bwd_kernel<T, L2_MAXJ>), dim3(nb), dim3(256), 0, s, npix, C, x, scale, dy, dx, ws); break;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 15) / 16), dim3(256), 0, s, ws, nb, C, dscale);
    HIP_OK(hipGetLastError());
}
void l2norm_bwd(int npix, int C, const float* x, const float* scale, const float* dy, float* dx, float* dscale, float* ws,
                hipStream_t s) {
    l2norm_bwd_t(npix, C, x, scale, dy, dx, dscale, ws, s);
}
void l2norm_bwd(int npix, int C, const bf16_t* x, const float* scale, const bf16_t* dy, bf16_t* dx, float* dscale, float* ws,
                hipStream_t s) {
    l2norm_bwd_t(npix, C, x, scale, dy, dx, dscale, ws, s);
}

// =================================================================================
// multibox heads: anchor a of image b lives in map i at (box type j, cell hw):
//   a = off[i] + j*hw[i] + cell ;  source row (b*hw[i] + cell) of buf[i], columns j*nvars..
// =================================================================================
constexpr int MAXV = 32;   // nvars <= 32  (num_classes <= 27)

// ---- work decomposition of the head kernels ---------------------------------------------------------------
// One workgroup owns HCH consecutive cells of one feature map of one image, i.e. the HCH * nj anchors whose
// head outputs are the HCH contiguous rows [cell][ld] of the fused head buffer.  Every global access of the
// kernels below is then a contiguous run: the rows themselves (HCH*ld floats), and per box type j the HCH
// anchors' [nvars] records of result / labels (anchor a = off + j*hw + cell).  Tiles are staged through LDS
// with 16-byte loads; thread (cell = t & 31, j = t >> 5) works on one anchor from LDS at odd strides.
constexpr int HCH = 32;
constexpr int HTHREADS = 256;               // 32 cells x 8 type slots (nj <= 8)
struct HeadGrid {
    int blk_off[MAX_MAPS + 1];              // first workgroup of each map
    int nchunk[MAX_MAPS];                   // cell chunks per image
    int ldp_max, nj_max;
};
static HeadGrid head_grid(const HeadLayout& L, int B) {
    HeadGrid g{};
    int off = 0;
    for (int i = 0; i < L.nmaps; ++i) {
        g.blk_off[i] = off;
        g.nchunk[i] = (L.hw[i] + HCH - 1) / HCH;
        off += g.nchunk[i] * B;
        g.ldp_max = std::max(g.ldp_max, L.ld[i] + 1);
        g.nj_max = std::max(g.nj_max, L.nj[i]);
    }
    for (int i = L.nmaps; i <= MAX_MAPS; ++i) g.blk_off[i] = off;
    SSD_REQUIRE(g.nj_max <= HTHREADS / HCH, "heads: at most %d box types per map", HTHREADS / HCH);
    return g;
}
struct HeadBlock {
    int map, b, cell0, ncell;
};
__device__ __forceinline__ HeadBlock head_block(const HeadLayout& L, const HeadGrid& G) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < MAX_MAPS; ++k)
        if (k < L.nmaps && (int)blockIdx.x >= G.blk_off[k]) i = k;
    const int local = blockIdx.x - G.blk_off[i];
    HeadBlock r;
    r.map = i;
    r.b = local / G.nchunk[i];
    r.cell0 = (local - r.b * G.nchunk[i]) * HCH;
    r.ncell = min(HCH, L.hw[i] - r.cell0);
    return r;
}

// contiguous global run (4-byte aligned) <-> LDS, 16-byte global accesses on the aligned middle
__device__ __forceinline__ void run_to_lds(float* dst, const float* __restrict__ src, int n) {
    const int pre = min(n, (int)((4u - (unsigned)(((size_t)src >> 2) & 3u)) & 3u));
    if ((int)threadIdx.x < pre) dst[threadIdx.x] = src[threadIdx.x];
    const int n4 = (n - pre) >> 2;
    for (int i = threadIdx.x; i < n4; i += HTHREADS) {
        const f32x4 v = ld4(src + pre + 4 * i);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[pre + 4 * i + e] = v[e];
    }
    for (int i = pre + 4 * n4 + threadIdx.x; i < n; i += HTHREADS) dst[i] = src[i];
}
__device__ __forceinline__ void lds_to_run(float* __restrict__ dst, const float* src, int n) {
    const int pre = min(n, (int)((4u - (unsigned)(((size_t)dst >> 2) & 3u)) & 3u));
    if ((int)threadIdx.x < pre) dst[threadIdx.x] = src[threadIdx.x];
    const int n4 = (n - pre) >> 2;
    for (int i = threadIdx.x; i < n4; i += HTHREADS) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = src[pre + 4 * i + e];
        st4(dst + pre + 4 * i, v);
    }
    for (int i = pre + 4 * n4 + threadIdx.x; i < n; i += HTHREADS) dst[i] = src[i];
}

template <bool TRAIN>
__global__ __launch_bounds__(HTHREADS) void heads_kernel(HeadLayout L, HeadGrid G, int B, float* __restrict__ result,
                                                         const float* __restrict__ labels, float* __restrict__ ce_out,
                                                         float* __restrict__ sl1_out, unsigned char* __restrict__ pos_out) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const HeadBlock k = head_block(L, G);
    const int nv = L.nvars, nc = nv - 4;
    const int ld = L.ld[k.map], ldp = ld + 1, nj = L.nj[k.map], hw = L.hw[k.map];
    float* hin = hsm;                         // [HCH][ldp]   raw head outputs, odd row pitch
    float* rec = hsm + HCH * G.ldp_max;       // [nj][HCH][nv] labels in, result out
    const size_t a0 = (size_t)k.b * L.A + L.off[k.map] + k.cell0;      // anchor of (type 0, first cell)
    const int nrec = k.ncell * nv;            // floats of one box type's run of [nv] records (<= 1024)
    // ---- every global load of the workgroup is issued before the first one is consumed ----------------------
    constexpr int HMAX = HCH * (MAXV * (HTHREADS / HCH) / 8 * 8) / 4 / HTHREADS + 1;     // float4 per thread, worst ld
    constexpr int RU = HCH * MAXV / HTHREADS;                                             // floats per thread and type
    const float* src = L.buf[k.map] + ((size_t)k.b * hw + k.cell0) * ld;                  // contiguous, 16-byte aligned rows
    const int n4 = (k.ncell * ld) >> 2;
    f32x4 hv[HMAX];
#pragma unroll
    for (int u = 0; u < HMAX; ++u) {
        const int i = threadIdx.x + HTHREADS * u;
        hv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < n4) hv[u] = ld4(src + 4 * i);
    }
    float lv[TRAIN ? HTHREADS / HCH : 1][RU];
    if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < HTHREADS / HCH; ++j) {
            const float* run = labels + (a0 + (size_t)j * hw) * nv;       // 4-byte aligned only: dword loads, 256 B per wave
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int i = threadIdx.x + HTHREADS * u;
                lv[j][u] = (j < nj && i < nrec) ? run[i] : 0.f;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < HMAX; ++u) {
        const int i = threadIdx.x + HTHREADS * u;
        if (i < n4) {
            const int row = (4 * i) / ld, col = 4 * i - row * ld;
#pragma unroll
            for (int e = 0; e < 4; ++e) hin[row * ldp + col + e] = hv[u][e];
        }
    }
    if constexpr (TRAIN) {
#pragma unroll
        for (int j = 0; j < HTHREADS / HCH; ++j)
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int i = threadIdx.x + HTHREADS * u;
                if (j < nj && i < nrec) rec[j * HCH * nv + i] = lv[j][u];
            }
    }
    __syncthreads();
    const int cell = threadIdx.x & (HCH - 1), j = threadIdx.x / HCH;
    if (cell < k.ncell && j < nj) {
        const float* zsrc = hin + cell * ldp + j * nv;
        float* r = rec + (j * HCH + cell) * nv;
        float z[MAXV], y[MAXV];
        float m = -__builtin_inff();
#pragma unroll
        for (int c = 0; c < MAXV; ++c) {
            z[c] = c < nv ? zsrc[c] : 0.f;
            y[c] = (TRAIN && c < nv) ? r[c] : 0.f;
            if (c < nc) m = fmaxf(m, z[c]);
        }
        // softmax = e_c / sum(e), e_c = exp(z_c - max) evaluated ONCE per class with the hardware exponential
        // (v_exp_f32 on (z - max) * log2 e: relative error ~ |z - max| * 2^-24 <= 1e-5 over the fp32 range, against
        // the 1e-3 bar); log once, accurately
        float e[MAXV];
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < MAXV; ++c) {
            e[c] = c < nc ? __expf(z[c] - m) : 0.f;
            se += e[c];
        }
        const float lse = m + logf(se);
        const float inv = 1.f / se;
#pragma unroll
        for (int c = 0; c < MAXV; ++c)
            if (c < nv) r[c] = c < nc ? e[c] * inv : z[c];
        if constexpr (TRAIN) {
            float ce = 0.f, sl = 0.f;
            const bool pos = y[nc - 1] == 0.f;
#pragma unroll
            for (int c = 0; c < MAXV; ++c) {
                if (c < nc) {
                    if (y[c] != 0.f) ce += y[c] * (lse - z[c]);
                } else if (c < nv) {
                    const float d = z[c] - y[c];
                    const float ad = fabsf(d);
                    sl += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
                }
            }
            const size_t idx = a0 + (size_t)j * hw + cell;
            ce_out[idx] = ce;
            sl1_out[idx] = pos ? sl : 0.f;
            pos_out[idx] = pos ? 1 : 0;
        }
    }
    __syncthreads();
    // result: per box type one contiguous run of [nv] records; dword stores, 256 B per wave
#pragma unroll
    for (int jj = 0; jj < HTHREADS / HCH; ++jj) {
        float* run = result + (a0 + (size_t)jj * hw) * nv;
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = threadIdx.x + HTHREADS * u;
            if (jj < nj && i < nrec) run[i] = rec[jj * HCH * nv + i];
        }
    }
}

static size_t heads_lds_bytes(const HeadLayout& L, const HeadGrid& G) {
    return ((size_t)HCH * G.ldp_max + (size_t)G.nj_max * HCH * L.nvars) * sizeof(float);
}

void heads_result(const HeadLayout& L, int B, float* result, hipStream_t s) {
    SSD_REQUIRE(L.nvars <= MAXV, "heads: num_classes + 5 must be <= %d", MAXV);
    const int total = B * L.A;
    const HeadGrid G = head_grid(L, B);
    ProfScope prof("heads_result", 0.0, 8.0 * total * L.nvars, s);
    hipLaunchKernelGGL(heads_kernel<false>, dim3(G.blk_off[MAX_MAPS]), dim3(HTHREADS), heads_lds_bytes(L, G), s, L, G, B, result,
                       nullptr, nullptr, nullptr, nullptr);
    HIP_OK(hipGetLastError());
}

// ---- per-sample: counts, sums, hard-negative selection (top-k by radix select) ------
// One workgroup per sample holds the sample's A cross-entropy values in registers (PT per thread): every
// pass of the MSB-first radix select (8 bits per pass over the float bit pattern, values are >= 0) runs from
// registers into an LDS histogram, the bin holding the k-th largest is found by a parallel suffix scan.
constexpr int SUMSQ_BLOCKS = 1024;
constexpr int LS_THREADS = 1024;
constexpr int LS_WAVES = LS_THREADS / 64;

__device__ __forceinline__ void block_sum3_1024(float& a, float& b, float& c, float* red) {     // red[3 * LS_WAVES]
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = a; red[LS_WAVES + (threadIdx.x >> 6)] = b; red[2 * LS_WAVES + (threadIdx.x >> 6)] = c;
    }
    __syncthreads();
    float ta = 0.f, tb = 0.f, tc = 0.f;
#pragma unroll
    for (int i = 0; i < LS_WAVES; ++i) { ta += red[i]; tb += red[LS_WAVES + i]; tc += red[2 * LS_WAVES + i]; }
    a = ta; b = tb; c = tc;
}

// the four losses from the per-sample results and the l2 partial sums: run by the LAST workgroup of loss_sample_kernel
// to finish (an integer ticket), always in the same order, so the result does not depend on which one that is
__device__ void loss_final_block(int B, float bnorm, const float* sample, const float* partial, int npartial, float wd,
                                 float* losses, double* dred) {
    double ss = 0.0;
    for (int i = threadIdx.x; i < npartial; i += LS_THREADS) ss += (double)__builtin_nontemporal_load(partial + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = ss;
    // the per-sample results: one load per thread (all in flight together), then summed in sample order
    __shared__ float s_conf[LS_THREADS / 2], s_loc[LS_THREADS / 2];
    if ((int)(threadIdx.x >> 1) < B)                        // B <= LS_THREADS / 2 (checked by the launcher)
        (threadIdx.x & 1 ? s_loc : s_conf)[threadIdx.x >> 1] = __builtin_nontemporal_load(sample + (threadIdx.x >> 1) * 4 + (threadIdx.x & 1));
    __syncthreads();
    if (threadIdx.x != 0) return;
    ss = 0.0;
    for (int i = 0; i < LS_WAVES; ++i) ss += dred[i];
    float conf = 0.f, loc = 0.f;
    for (int b = 0; b < B; ++b) {
        conf += s_conf[b];
        loc += s_loc[b];
    }
    conf /= bnorm;
    loc /= bnorm;
    const float l2 = wd * (float)(0.5 * ss);
    losses[0] = conf + loc + l2;
    losses[1] = loc;
    losses[2] = conf;
    losses[3] = l2;
    __threadfence_system();      // the executor's buffer is host memory
}

template <int PT>
__device__ __forceinline__ void loss_sample_body(int b, int A, float bnorm, const float* __restrict__ ce,
                                                 const float* __restrict__ sl1, const unsigned char* __restrict__ pos,
                                                 unsigned char* __restrict__ sel, float* __restrict__ sample) {
    __shared__ float red[3 * LS_WAVES];
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_k, sh_ties, sh_wsum[4], sh_wtot[LS_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* cb = ce + (size_t)b * A;
    const float* lb = sl1 + (size_t)b * A;
    const unsigned char* pb = pos + (size_t)b * A;
    unsigned char* sb = sel + (size_t)b * A;

    // the sample's values, loaded once (coalesced, all loads in flight together)
    unsigned u[PT];          // negatives[a] = pos ? 0 : ce  (ssdvgg.py:459) as bit patterns
    unsigned pmask = 0, vmask = 0;
    float npos = 0.f, psum = 0.f, lsum = 0.f;
    {   // pass 1 issues every load, pass 2 consumes: a consumer between two loads would make each load wait in turn
        float cv[PT], lv[PT];
        unsigned char pv[PT];
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int a = tid + LS_THREADS * r;
            const int ac = a < A ? a : 0;           // unconditional loads from a clamped index (a select on the loaded
            cv[r] = cb[ac];                         // value would be a consumer again); masked in pass 2
            lv[r] = lb[ac];
            pv[r] = pb[ac];
        }
#pragma unroll
        for (int r = 0; r < PT; ++r) {
            const int a = tid + LS_THREADS * r;
            u[r] = 0u;
            if (a < A) {
                const bool p = pv[r] != 0;
                lsum += lv[r];
                vmask |= 1u << r;
                if (p) { pmask |= 1u << r; npos += 1.f; psum += cv[r]; }
                else u[r] = __float_as_uint(cv[r]);
            }
        }
    }
    block_sum3_1024(npos, psum, lsum, red);
    const int pos_n = (int)npos;
    const int neg_n = A - pos_n;
    const int k = min(neg_n, 3 * pos_n);
    if (pos_n == 0) {   // ssdvgg.py:513-516,552-555: the sample contributes exactly 0
#pragma unroll
        for (int r = 0; r < PT; ++r)
            if (vmask >> r & 1u) sb[tid + LS_THREADS * r] = 0;
        if (tid == 0) {
            sample[b * 4 + 0] = 0.f; sample[b * 4 + 1] = 0.f; sample[b * 4 + 2] = 0.f; sample[b * 4 + 3] = 0.f;
        }
        return;
    }
    unsigned prefix = 0, kk = (unsigned)k;   // kk-th largest among entries matching prefix
    bool ties = false;
    if (k > 0) {
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned himask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
#pragma unroll
            for (int r = 0; r < PT; ++r) {
                const bool in = (vmask >> r & 1u) && (u[r] & himask) == (prefix & himask);
                const unsigned digit = (u[r] >> shift) & 255u;
                const unsigned long long bal = __ballot(in);
                if (bal == 0ull) continue;
                // a wave whose entries all fall into one bin (the usual case for the exponent byte) adds once
                const unsigned d0 = __builtin_amdgcn_readlane(digit, __ffsll((long long)bal) - 1);
                if (__ballot(in && digit == d0) == bal) {
                    if (lane == __ffsll((long long)bal) - 1) atomicAdd(&hist[d0], (unsigned)__popcll(bal));
                } else if (in) {
                    atomicAdd(&hist[digit], 1u);
                }
            }
            __syncthreads();
            // bin of the kk-th largest: inclusive suffix sums over bins 255..0 by the first four waves
            unsigned h = 0, inc = 0;
            if (tid < 256) {
                h = hist[255 - tid];
                inc = h;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(inc, o, 64);
                    if (lane >= o) inc += t;
                }
                if (lane == 63) sh_wsum[wv] = inc;
            }
            __syncthreads();
            if (tid < 256) {
                for (int i = 0; i < wv; ++i) inc += sh_wsum[i];
                const unsigned exc = inc - h;
                if (inc >= kk && exc < kk) {
                    sh_prefix = prefix | ((unsigned)(255 - tid) << shift);
                    sh_k = kk - exc;
                    sh_ties = h > kk - exc ? 1u : 0u;
                }
            }
            __syncthreads();
            prefix = sh_prefix;
            kk = sh_k;
            ties = sh_ties != 0u;          // meaningful after the last pass: more entries equal T than are taken
        }
    }
    // prefix = bit pattern of the threshold T; kk = how many entries == T are taken
    const float T = __uint_as_float(prefix);
    float tsum = 0.f, z0 = 0.f, z1 = 0.f;
    if (k > 0) {
#pragma unroll
        for (int r = 0; r < PT; ++r)
            if ((vmask >> r & 1u) && u[r] > prefix) tsum += __uint_as_float(u[r]);
    }
    block_sum3_1024(tsum, z0, z1, red);
    if (k > 0) tsum += (float)kk * T;
    // selection mask; ties at T go to the LOWER index first (tf.nn.top_k)
    if (!ties) {
#pragma unroll
        for (int r = 0; r < PT; ++r)
            if (vmask >> r & 1u) sb[tid + LS_THREADS * r] = ((pmask >> r & 1u) || (k > 0 && u[r] >= prefix)) ? 1 : 0;
    } else {
        unsigned base = 0;
#pragma unroll
        for (int r = 0; r < PT; ++r) {          // a = tid + 1024 r ascends with (r, tid)
            const bool in = vmask >> r & 1u;
            const bool eq = in && u[r] == prefix;      // positives hold 0: they tie with T == 0 like the reference's zeros
            const unsigned long long bal = __ballot(eq);
            const unsigned before = __popcll(bal & ((1ull << lane) - 1ull));
            __syncthreads();
            if (lane == 0) sh_wtot[wv] = __popcll(bal);
            __syncthreads();
            unsigned wbase = base, tot = 0;
#pragma unroll
            for (int i = 0; i < LS_WAVES; ++i) {
                if (i < wv) wbase += sh_wtot[i];
                tot += sh_wtot[i];
            }
            const bool take = u[r] > prefix || (eq && wbase + before < kk);
            if (in) sb[tid + LS_THREADS * r] = ((pmask >> r & 1u) || take) ? 1 : 0;
            base += tot;
        }
    }
    if (tid == 0) {
        sample[b * 4 + 0] = (psum + tsum) / (float)pos_n;
        sample[b * 4 + 1] = lsum / (float)pos_n;
        sample[b * 4 + 2] = 1.f / ((float)pos_n * bnorm);
        sample[b * 4 + 3] = (float)pos_n;
    }
}

// One workgroup per sample.  A step may run the samples in several launches (forward lanes, net.hip): b_off is this launch's
// first sample, B the step's total -- the workgroup that draws ticket B - 1, whichever launch it belongs to, finishes.
template <int PT>
__global__ __launch_bounds__(LS_THREADS) void loss_sample_kernel(int B, int b_off, int A, float bnorm, const float* __restrict__ ce,
                                                                 const float* __restrict__ sl1,
                                                                 const unsigned char* __restrict__ pos,
                                                                 unsigned char* __restrict__ sel,
                                                                 float* __restrict__ sample, const float* __restrict__ partial,
                                                                 int npartial, float wd, float* __restrict__ losses,
                                                                 unsigned* __restrict__ ticket) {
    __shared__ double dred[LS_WAVES];
    __shared__ unsigned sh_last;
    loss_sample_body<PT>((int)blockIdx.x + b_off, A, bnorm, ce, sl1, pos, sel, sample);
    // the workgroup that draws the last ticket reduces everything (loss_final_block); the ticket resets itself
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                   // this sample's results before the ticket
        const unsigned t = atomicAdd(ticket, 1u);
        sh_last = t == (unsigned)B - 1u ? 1u : 0u;
        if (sh_last) *ticket = 0u;
    }
    __syncthreads();
    if (!sh_last) return;
    __threadfence();
    loss_final_block(B, bnorm, sample, partial, npartial, wd, losses, dred);
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ w, size_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = ld4(w + i * 4);
        s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}


size_t loss_work_bytes(int B, int A) {
    const size_t n = (size_t)B * A;
    size_t bytes = 0;
    bytes += n * 4 * 2;                       // ce, sl1
    bytes += ((n + 15) / 16) * 16 * 2;        // pos, sel
    bytes += (size_t)B * 4 * 4 + SUMSQ_BLOCKS * 4 + 64 + 64;
    return bytes + 256;
}

void loss_work_carve(LossWork& w, void* base, int B, int A) {
    const size_t n = (size_t)B * A;
    char* p = (char*)base;
    w.ce = (float*)p; p += n * 4;
    w.sl1 = (float*)p; p += n * 4;
    w.pos = (unsigned char*)p; p += ((n + 15) / 16) * 16;
    w.sel = (unsigned char*)p; p += ((n + 15) / 16) * 16;
    w.sample = (float*)p; p += (size_t)B * 4 * 4;
    w.partial = (float*)p; p += SUMSQ_BLOCKS * 4;
    w.losses = (float*)p; p += 64;
    w.ticket = (unsigned*)p;          // zero at allocation, resets itself
}

void l2_partials(const float* filters, size_t nfilters, LossWork& w, hipStream_t s) {
    ProfScope prof("l2_partials", 0.0, 4.0 * nfilters, s);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, filters, nfilters, w.partial);
    HIP_OK(hipGetLastError());
}

void multibox_loss(const HeadLayout& L, int B, int b_off, int B_total, const float* result, const float* labels, LossWork& w,
                   float weight_decay, float bnorm, hipStream_t s) {
    SSD_REQUIRE(L.nvars <= MAXV, "heads: num_classes + 5 must be <= %d", MAXV);
    SSD_REQUIRE(L.A <= 32 * LS_THREADS, "loss: at most %d anchors", 32 * LS_THREADS);
    SSD_REQUIRE(B_total <= LS_THREADS / 2, "loss: at most %d images per step", LS_THREADS / 2);
    const int total = B * L.A;
    const HeadGrid G = head_grid(L, B);
    if (!(bnorm > 0.f)) bnorm = (float)B_total;          // reduce_mean over this step's own batch (ssdvgg.py:520,559)
    const size_t o = (size_t)b_off * L.A;                // this launch's slice of the per-anchor work arrays
    ProfScope prof("multibox_loss", 0.0, 12.0 * total * L.nvars, s);
    hipLaunchKernelGGL(heads_kernel<true>, dim3(G.blk_off[MAX_MAPS]), dim3(HTHREADS), heads_lds_bytes(L, G), s, L, G, B,
                       const_cast<float*>(result), labels, w.ce + o, w.sl1 + o, w.pos + o);
    const int pt = (L.A + LS_THREADS - 1) / LS_THREADS;
    if (pt <= 9)
        hipLaunchKernelGGL(loss_sample_kernel<9>, dim3(B), dim3(LS_THREADS), 0, s, B_total, b_off, L.A, bnorm, w.ce, w.sl1, w.pos, w.sel, w.sample,
                           w.partial, SUMSQ_BLOCKS, weight_decay, w.losses, w.ticket);
    else if (pt <= 24)
        hipLaunchKernelGGL(loss_sample_kernel<24>, dim3(B), dim3(LS_THREADS), 0, s, B_total, b_off, L.A, bnorm, w.ce, w.sl1, w.pos, w.sel, w.sample,
                           w.partial, SUMSQ_BLOCKS, weight_decay, w.losses, w.ticket);
    else
        hipLaunchKernelGGL(loss_sample_kernel<32>, dim3(B), dim3(LS_THREADS), 0, s, B_total, b_off, L.A, bnorm, w.ce, w.sl1, w.pos, w.sel, w.sample,
                           w.partial, SUMSQ_BLOCKS, weight_decay, w.losses, w.ticket);
    HIP_OK(hipGetLastError());
}

// d/d(logits) = sel * (softmax - labels) * w_b ; d/d(loc) = pos * clip(loc - gt, -1, 1) * w_b,
// w_b = 1 / (pos_n_b * B)  (reduce_mean over the batch of per-sample normalised sums).
// All but a few hundred anchors per image are neither selected nor positive: a workgroup zero-fills its
// [HCH][ld] tile of the gradient buffer in LDS, the few flagged anchors fetch their result / label records and
// fill their columns, and the tile leaves as one contiguous run of 16-byte stores (pad columns included).
template <typename T>
__global__ __launch_bounds__(HTHREADS) void loss_grad_kernel(HeadLayout L, HeadGrid G, int B, int b_off, const float* __restrict__ result,
                                                             const float* __restrict__ labels,
                                                             const unsigned char* __restrict__ pos,
                                                             const unsigned char* __restrict__ sel,
                                                             const float* __restrict__ sample) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];      // [HCH][ld]
    const HeadBlock k = head_block(L, G);
    const int nv = L.nvars, nc = nv - 4;
    const int ld = L.ld[k.map], nj = L.nj[k.map], hw = L.hw[k.map];
    const int n4 = (k.ncell * ld) >> 2;
    for (int i = threadIdx.x; i < n4; i += HTHREADS) *reinterpret_cast<f32x4*>(gsm + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int cell = threadIdx.x & (HCH - 1), j = threadIdx.x / HCH;
    if (cell < k.ncell && j < nj) {
        const size_t idx = (size_t)k.b * L.A + L.off[k.map] + (size_t)j * hw + k.cell0 + cell;
        const bool isel = sel[idx], ipos = pos[idx];
        if (isel || ipos) {
            const float wb = sample[(k.b + b_off) * 4 + 2];
            const float* r = result + idx * nv;
            const float* y = labels + idx * nv;
            float* dst = gsm + cell * ld + j * nv;
            if (isel)
                for (int c = 0; c < nc; ++c) dst[c] = (r[c] - y[c]) * wb;
            if (ipos)
                for (int c = nc; c < nv; ++c) dst[c] = fminf(fmaxf(r[c] - y[c], -1.f), 1.f) * wb;
        }
    }
    __syncthreads();
    T* out = static_cast<T*>(L.dbuf[k.map]) + ((size_t)k.b * hw + k.cell0) * ld;
    for (int i = threadIdx.x; i < n4; i += HTHREADS) st4t(out + 4 * i, *reinterpret_cast<const f32x4*>(gsm + 4 * i));
}

void multibox_loss_grad(const HeadLayout& L, int B, int b_off, const float* result, const float* labels, const LossWork& w,
                        hipStream_t s) {
    const int total = B * L.A;
    const HeadGrid G = head_grid(L, B);
    const size_t lds = (size_t)HCH * (G.ldp_max - 1) * sizeof(float);
    const size_t o = (size_t)b_off * L.A;
    ProfScope prof("multibox_loss_grad", 0.0, (L.grad_bf16 ? 10.0 : 12.0) * total * L.nvars, s);
    if (L.grad_bf16)
        hipLaunchKernelGGL(loss_grad_kernel<bf16_t>, dim3(G.blk_off[MAX_MAPS]), dim3(HTHREADS), lds, s, L, G, B, b_off, result, labels,
                           w.pos + o, w.sel + o, w.sample);
    else
        hipLaunchKernelGGL(loss_grad_kernel<float>, dim3(G.blk_off[MAX_MAPS]), dim3(HTHREADS), lds, s, L, G, B, b_off, result, labels,
                           w.pos + o, w.sel + o, w.sample);
    HIP_OK(hipGetLastError());
}

// =================================================================================
// optimizer
// =================================================================================
__global__ __launch_bounds__(256) void momentum_kernel(float* __restrict__ w, float* __restrict__ acc,
                                                       const float* __restrict__ g, size_t n4, float lr, float mom,
                                                       float gscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = mom * ld4(acc + i * 4) + gscale * ld4(g + i * 4);
        st4(acc + i * 4, a);
        st4(w + i * 4, ld4(w + i * 4) - lr * a);
    }
}

void momentum_update(float* w, float* acc, const float* g, size_t n, float lr, float momentum, float gscale,
                     hipStream_t s) {
    SSD_REQUIRE(n % 4 == 0, "momentum: arena size must be a multiple of 4");
    ProfScope prof("momentum_update", 0.0, 20.0 * n, s);
    hipLaunchKernelGGL(momentum_kernel, dim3(grid_for(n / 4, 256, 256 * 16)), dim3(256), 0, s, w, acc, g, n / 4, lr,
                       momentum, gscale);
    HIP_OK(hipGetLastError());
}

// Gradient range <-> bf16 message buffer of the data-parallel all-reduce (parallel.py, allreduce_dtype = 'bf16'): the payload
// that crosses xGMI is halved; masters, momentum and the local gradient arena stay fp32.  Round-to-nearest-even like every other
// bf16 store of the library; n need not be a multiple of anything.
__global__ __launch_bounds__(256) void grads_to_bf16_kernel(const float* __restrict__ g, unsigned short* __restrict__ out, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = ld4(g + i * 4);
        *reinterpret_cast<u32x2*>(out + i * 4) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) out[n4 * 4 + threadIdx.x] = f2bf(g[n4 * 4 + threadIdx.x]);
}
__global__ __launch_bounds__(256) void grads_from_bf16_kernel(const unsigned short* __restrict__ in, float* __restrict__ g, size_t n) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(in + i * 4);
        st4(g + i * 4, f32x4{lo2f(w[0]), hi2f(w[0]), lo2f(w[1]), hi2f(w[1])});
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) g[n4 * 4 + threadIdx.x] = lo2f((unsigned)in[n4 * 4 + threadIdx.x]);
}
void grads_to_bf16(const float* g, void* out, size_t n, hipStream_t s) {
    SSD_REQUIRE(((uintptr_t)g % 16 == 0) && ((uintptr_t)out % 8 == 0), "gradient range / message buffer must be 16- / 8-byte aligned");
    if (n == 0) return;
    hipLaunchKernelGGL(grads_to_bf16_kernel, dim3(grid_for(n / 4 + 1, 256, 256 * 16)), dim3(256), 0, s, g, (unsigned short*)out, n);
    HIP_OK(hipGetLastError());
}
void grads_from_bf16(const void* in, float* g, size_t n, hipStream_t s) {
    SSD_REQUIRE(((uintptr_t)g % 16 == 0) && ((uintptr_t)in % 8 == 0), "gradient range / message buffer must be 16- / 8-byte aligned");
    if (n == 0) return;
    hipLaunchKernelGGL(grads_from_bf16_kernel, dim3(grid_for(n / 4 + 1, 256, 256 * 16)), dim3(256), 0, s, (const unsigned short*)in, g, n);
    HIP_OK(hipGetLastError());
}

// gradient arena of a step without samples: d(l2_loss)/dw = wd * w on the filters, zero elsewhere
__global__ __launch_bounds__(256) void null_grads_kernel(const float* __restrict__ w, float* __restrict__ g, size_t nf4, size_t n4,
                                                         float wd) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        st4(g + i * 4, i < nf4 ? wd * ld4(w + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f});
}

void null_gradients(const float* w, float* g, size_t nfilters, size_t n, float wd, hipStream_t s) {
    SSD_REQUIRE(n % 4 == 0 && nfilters % 4 == 0, "arena sizes must be multiples of 4");
    hipLaunchKernelGGL(null_grads_kernel, dim3(grid_for(n / 4, 256, 256 * 16)), dim3(256), 0, s, w, g, nfilters / 4, n / 4, wd);
    HIP_OK(hipGetLastError());
}

// Shader-clock monitor (measurement aid, tools/power_probe.py): ONE wave spins next to whatever else runs on the GPU and
// records, every `period` ticks of the constant 100 MHz counter (s_memrealtime), how many shader-clock cycles
// (s_memtime) went by: out[i] = cycles per period -> MHz = out[i] / period * 100.
__global__ void clock_monitor_kernel(unsigned* __restrict__ out, int nsamples, unsigned period) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < nsamples; ++i) {
        const unsigned long long w0 = wall_clock64();
        const unsigned long long c0 = clock64();
        unsigned long long w1;
        do {
            __builtin_amdgcn_s_sleep(32);
            w1 = wall_clock64();
        } while (w1 - w0 < period);
        const unsigned long long c1 = clock64();
        // normalise to exactly `period` wall ticks
        out[i] = (unsigned)((double)(c1 - c0) * (double)period / (double)(w1 - w0));
    }
}

void clock_monitor(unsigned* out, int nsamples, unsigned period, hipStream_t s) {
    hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, s, out, nsamples, period);
    HIP_OK(hipGetLastError());
}

void fill_zero(void* p, size_t bytes, hipStream_t s) { HIP_OK(hipMemsetAsync(p, 0, bytes, s)); }

}  // namespace ssd
