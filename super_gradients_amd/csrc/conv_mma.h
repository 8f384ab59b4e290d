// bf16x3 arithmetic helpers shared by the convolution kernels (conv.hip, wgrad_patch.hip): the fp32 -> three bf16 pieces split, the bf16 MFMA
// and gfx950's LDS transpose read - each with its host-emulation twin (SGX_EMU, tests only).
#pragma once
#include "sgx_common.h"

__device__ __forceinline__ unsigned sgx_f2u(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
__device__ __forceinline__ float sgx_u2f(unsigned u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// (bf16(b) << 16) | bf16(a), round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned sgx_pack_bf16(float a, float b) {
#ifdef SGX_EMU
    const unsigned ua = sgx_f2u(a), ub = sgx_f2u(b);
    const unsigned ra = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16, rb = (ub + 0x7fffu + ((ub >> 16) & 1u)) >> 16;
    return (ra & 0xffffu) | (rb << 16);
#else
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
#endif
}
// x = hi + mid + lo with every piece rounded to nearest: |x - (hi + mid + lo)| <= 2^-27 |x|, the residuals are exact in fp32, and the
// pieces carry mixed signs, so the cross terms the six-product scheme drops (mid*lo, lo*mid, lo*lo: <= 2^-26 of a product) are unbiased.
// (A truncating split is one instruction cheaper per pair but leaves every dropped term with the sign of the product: measured as a
// 2x larger end-to-end error than the fp32 matrix pipe on the YOLO-NAS-M golden fixture.)
__device__ __forceinline__ void sgx_split3(const float4& v, uint2& h, uint2& m, uint2& l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned hp[2], mp[2], lp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        hp[i] = sgx_pack_bf16(a, b);
        const float ra = a - sgx_u2f(hp[i] << 16), rb = b - sgx_u2f(hp[i] & 0xffff0000u);
        mp[i] = sgx_pack_bf16(ra, rb);
        const float sa = ra - sgx_u2f(mp[i] << 16), sb = rb - sgx_u2f(mp[i] & 0xffff0000u);
        lp[i] = sgx_pack_bf16(sa, sb);
    }
    h = make_uint2(hp[0], hp[1]);
    m = make_uint2(mp[0], mp[1]);
    l = make_uint2(lp[0], lp[1]);
}
#ifdef SGX_EMU
// host emulation of v_mfma_f32_32x32x16_bf16: lane l holds A[row l%32][k = 8*(l/32) .. +7] and B[k = 8*(l/32) .. +7][col l%32]
static inline sgx_f32x16 sgx_mfma_bf16(const uint4& a, const uint4& b, sgx_f32x16 c) {
    uint64_t u[4];
    memcpy(&u[0], &a, 16);
    memcpy(&u[2], &b, 16);
    auto x = sgx_emu::xchg_put(u, 4);
    const int l = sgx_emu::t_lane;
    const int col = l & 31;
    sgx_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        for (int half = 0; half < 2; ++half) {
            unsigned short av[8], bv[8];
            memcpy(av, &x.w->xbuf[x.buf][row + 32 * half][0], 16);
            memcpy(bv, &x.w->xbuf[x.buf][col + 32 * half][2], 16);
            for (int k = 0; k < 8; ++k) acc = fmaf(sgx_u2f((unsigned)av[k] << 16), sgx_u2f((unsigned)bv[k] << 16), acc);
        }
        d[r] = acc;
    }
    return d;
}
#else
typedef __bf16 sgx_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ sgx_f32x16 sgx_mfma_bf16(const uint4& a, const uint4& b, sgx_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sgx_bf16x8, a), __builtin_bit_cast(sgx_bf16x8, b), c, 0, 0, 0);
}
#endif

// ds_read_b64_tr_b16 (gfx950's LDS transpose read): every lane reads the four 16-bit elements at its own 8-byte-aligned LDS address; inside
// each group of 16 lanes the 16 x 4 matrix In[lane][e] comes back transposed in 4 x 4 blocks - lane i receives In[(i >> 2) + 4 j][i & 3],
// j = 0..3 (measured on the chip: tools/probe_tr_read.hip, profiles/r3zi_probe_ds_read_tr_b16.txt).  If the 16 lanes address a
// [4 rows][16 columns] block of a row-major bf16 image (lane q: row q >> 2, columns 4 (q & 3) .. + 3), lane i gets column i of the four
// rows: an MFMA operand whose reduction index runs along the ROWS of the LDS image (pixels of a weight gradient) without a transposing
// store.  All 64 lanes take part (the host emulation exchanges through the wave buffer).
#ifdef SGX_EMU
static inline uint2 sgx_lds_tr_read(const unsigned short* p) {
    uint64_t u;
    memcpy(&u, p, 8);
    auto x = sgx_emu::xchg_put(&u, 1);
    const int l = sgx_emu::t_lane, g = l & ~15, i = l & 15;
    unsigned short o[4];
    for (int j = 0; j < 4; ++j) {
        unsigned short in[4];
        memcpy(in, &x.w->xbuf[x.buf][g + (i >> 2) + 4 * j][0], 8);
        o[j] = in[i & 3];
    }
    uint2 r;
    memcpy(&r, o, 8);
    return r;
}
#else
typedef short sgx_i16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 sgx_lds_tr_read(const unsigned short* p) {
    const sgx_i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sgx_i16x4*)(p));
    return __builtin_bit_cast(uint2, v);
}
#endif
