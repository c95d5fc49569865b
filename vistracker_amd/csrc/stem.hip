// stem.hip -- the 7 x 7 / stride 2 / pad 3 convolution at the head of the HGFilter encoders (model/HGFilters.py:118-130: conv1 = Conv2d(in_ch, 64, 7, 2, 3)
// with bias; 64 output channels in the image encoder, 32 in the triplane encoder).  Cin is 5 (image encoder: RGB + two masks)
// or 1 (triplane encoder: one rendered mask), so K = 49 Cin is 245 / 49:
// too thin for the split-f16 implicit GEMM of conv.hip (K chunks of 32 channels) and small in absolute terms (53 GFLOP per 16 frames, 1 % of the pass),
// but until round 6 it was the one convolution the encoder handed to MIOpen -- with a layout transpose either side, a kernel search at the first call of
// every process and a second backend nobody measured.  Here: plain fp32 FMAs (the arithmetic of the reference), weights as SCALAR operands.
//   * workgroup = 16 x 16 output pixels of one frame, 4 waves; a lane owns one pixel and all CO = 64 / 32 output channels (CO accumulators);
//   * the 37 x 37 x Cin input patch is loaded once into LDS (zero padded); per (channel, ky, kx) a lane reads ONE value from LDS and issues 64 FMAs
//     whose weight operand is an SGPR: the weights are stored [Cin][7][7][64] so that the 64 weights of a tap are 256 contiguous bytes that the scalar
//     unit fetches (s_load_dwordx16) for the whole wave -- no vector-memory or LDS traffic for the larger operand at all;
//   * epilogue through LDS (per wave [64 pixels][33]) so that the NHWC result leaves as 128 contiguous bytes per 8 lanes.
// Bound: VALU issue (245 x 64 FMAs per pixel, 39 T FMA/s per chip un-packed).
#include "common.h"

#define ST_T 16                     /* output tile edge */
#define ST_P (2 * ST_T + 5)         /* input patch edge: 37 */
#define ST_PP 40                    /* patch row pitch (floats) */
#define ST_SP 33                    /* staging pitch per pixel (floats): conflict-free transposition */

struct vt_stem7x7 {
    float *w;       // [Cin][7][7][Cout]
    float *bias;    // [Cout] (zeros without one)
    int cin, cout;
};

template <int ST_CO>
__global__ __launch_bounds__(256) void stem7x7_kernel(const float *__restrict__ in, int in_cstride, int in_coff, int Cin, int H, int W, const float *__restrict__ wT,
                                                      const float *__restrict__ bias, float *__restrict__ out, int out_cstride, int out_coff, int H2, int W2)
{
    extern __shared__ float smem[];         // patch [Cin][37][40], afterwards the staging tiles [4 waves][64][33]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tiles_x = (W2 + ST_T - 1) / ST_T, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
    const int iy0 = 2 * ST_T * ty - 3, ix0 = 2 * ST_T * tx - 3;
    const float *__restrict__ inb = in + (size_t)b * H * W * in_cstride + in_coff;
    for (int p = tid; p < ST_P * ST_P; p += 256) {
        const int py = p / ST_P, px = p - py * ST_P, iy = iy0 + py, ix = ix0 + px;
        const bool inside = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const float *src = inb + ((size_t)(inside ? iy : 0) * W + (inside ? ix : 0)) * in_cstride;
        for (int ci = 0; ci < Cin; ci++) smem[(ci * ST_P + py) * ST_PP + px] = inside ? src[ci] : 0.f;
    }
    __syncthreads();
    const int oyl = 4 * wave + (lane >> 4), oxl = lane & 15;
    float acc[ST_CO];
#pragma unroll
    for (int co = 0; co < ST_CO; co++) acc[co] = 0.f;
    for (int ci = 0; ci < Cin; ci++) {
        for (int ky = 0; ky < 7; ky++) {
            const float *xr = smem + (ci * ST_P + 2 * oyl + ky) * ST_PP + 2 * oxl;
            const float *__restrict__ wr = wT + (size_t)((ci * 7 + ky) * 7) * ST_CO;        // wave-uniform: scalar loads
#pragma unroll
            for (int kx = 0; kx < 7; kx++) {
                const float x = xr[kx];
#pragma unroll
                for (int co = 0; co < ST_CO; co++) acc[co] = __builtin_fmaf(x, wr[kx * ST_CO + co], acc[co]);
            }
        }
    }
    __syncthreads();                        // every wave is done with the patch
    float *st = smem + wave * (64 * ST_SP);
    const int oy0 = ST_T * ty + 4 * wave, ox0 = ST_T * tx;
#pragma unroll
    for (int h = 0; h < ST_CO / 32; h++) {
        if (h) __syncthreads();
#pragma unroll
        for (int k = 0; k < 32; k++) st[lane * ST_SP + k] = acc[32 * h + k] + bias[32 * h + k];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int item = r * 64 + lane, px = item >> 3, piece = item & 7;
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
            const float *s = st + px * ST_SP + piece * 4;
            const float4 v = make_float4(s[0], s[1], s[2], s[3]);
            if (oy < H2 && ox < W2) *reinterpret_cast<float4 *>(out + ((size_t)(b * H2 + oy) * W2 + ox) * out_cstride + out_coff + 32 * h + piece * 4) = v;
        }
    }
}

extern "C" int vt_stem7x7_create(vt_stem7x7 **out, const float *weight, const float *bias, int cout, int cin, void *stream)
{
    VT_REQUIRE(out && weight && (cout == 64 || cout == 32) && cin >= 1 && cin <= 8, "vt_stem7x7_create: needs Cout in {32, 64} and Cin in 1..8");
    const int ST_CO = cout;
    const size_t n = (size_t)cin * 49 * ST_CO;
    float *host = new float[n + ST_CO];
    for (int co = 0; co < ST_CO; co++)
        for (int ci = 0; ci < cin; ci++)
            for (int t = 0; t < 49; t++) host[((size_t)ci * 49 + t) * ST_CO + co] = weight[((size_t)co * cin + ci) * 49 + t];      // (Cout, Cin, 7, 7) -> [Cin][7][7][Cout]
    for (int co = 0; co < ST_CO; co++) host[n + co] = bias ? bias[co] : 0.f;
    vt_stem7x7 *h = new vt_stem7x7();
    h->cin = cin; h->cout = cout; h->w = nullptr;
    const hipError_t e = hipMalloc(reinterpret_cast<void **>(&h->w), (n + ST_CO) * sizeof(float));
    if (e != hipSuccess) { delete[] host; delete h; VT_HIP(e); }
    h->bias = h->w + n;
    const hipError_t e2 = hipMemcpyAsync(h->w, host, (n + ST_CO) * sizeof(float), hipMemcpyHostToDevice, vt_stream(stream));
    const hipError_t e3 = e2 == hipSuccess ? hipStreamSynchronize(vt_stream(stream)) : e2;
    delete[] host;
    if (e3 != hipSuccess) { (void)hipFree(h->w); delete h; VT_HIP(e3); }
    *out = h;
    return VT_OK;
}
extern "C" void vt_stem7x7_destroy(vt_stem7x7 *h) { if (!h) return; (void)hipFree(h->w); delete h; }

extern "C" int vt_stem7x7_forward(const vt_stem7x7 *h, const float *in, int in_cstride, int in_coff, int B, int H, int W, float *out, int out_cstride, int out_coff, void *stream)
{
    VT_REQUIRE(h && in && out && B > 0 && H > 0 && W > 0, "vt_stem7x7_forward: bad arguments");
    VT_REQUIRE(in_coff >= 0 && in_coff + h->cin <= in_cstride, "vt_stem7x7_forward: input channel slice [%d, %d) outside %d channels", in_coff, in_coff + h->cin, in_cstride);
    VT_REQUIRE(out_coff >= 0 && out_coff + h->cout <= out_cstride && out_cstride % 4 == 0 && out_coff % 4 == 0, "vt_stem7x7_forward: output channel slice must be 16-byte aligned inside the NHWC tensor");
    const int H2 = (H - 1) / 2 + 1, W2 = (W - 1) / 2 + 1;           // floor((H + 2 * 3 - 7) / 2) + 1
    const size_t patch = (size_t)h->cin * ST_P * ST_PP * sizeof(float), stage = (size_t)4 * 64 * ST_SP * sizeof(float);
    const size_t lds = patch > stage ? patch : stage;
    const dim3 grid(((H2 + ST_T - 1) / ST_T) * ((W2 + ST_T - 1) / ST_T), B);
    if (h->cout == 64) hipLaunchKernelGGL(stem7x7_kernel<64>, grid, dim3(256), lds, vt_stream(stream), in, in_cstride, in_coff, h->cin, H, W, h->w, h->bias, out, out_cstride, out_coff, H2, W2);
    else hipLaunchKernelGGL(stem7x7_kernel<32>, grid, dim3(256), lds, vt_stream(stream), in, in_cstride, in_coff, h->cin, H, W, h->w, h->bias, out, out_cstride, out_coff, H2, W2);
    VT_LAUNCH_CHECK();
    return VT_OK;
}
