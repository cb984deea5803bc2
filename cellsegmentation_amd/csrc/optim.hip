// Adam over a list of fp32 tensors in ONE launch (round 3).  The reference's drivers build torch.optim.Adam (train_tile.py:282,
// train_image.py:476, train_seg.py:309); torch's fused implementation of that update is 5 multi_tensor_apply launches at 2.9 TB/s
// on the ResNet-50 tile step (658 MB of parameter / gradient / moment traffic, 0.23 ms of a 7.4 ms step).  Here: a device table of
// (p, m, v, n), the gradients' addresses BY VALUE in the kernel arguments (they change every step: autograd hands out fresh
// tensors), a device list of 16 Ki-element chunks, one workgroup per chunk, 16 bytes per lane.
//   g' = g + weight_decay * p            (torch.optim.Adam's L2 form, not AdamW)
//   m  = m + (1 - beta1) * (g' - m)      (lerp, as torch)
//   v  = beta2 * v + (1 - beta2) * g'^2
//   p  = p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include "cs_common.h"
#include <math.h>

namespace {

constexpr int kAdamChunk = 16384;
constexpr int kAdamMaxTensors = 320;

struct AdamGrads { const float* g[kAdamMaxTensors]; };

// (omb1 = 1 - beta1 and omb2 = 1 - beta2 arrive rounded from DOUBLE, as torch forms them: 1.f - 0.999f is off by 1.3e-5 relative)
__device__ __forceinline__ void adam1(float& p, float& m, float& v, float g, float step_size, float inv_sqrt_bc2, float omb1, float beta2,
                                      float omb2, float eps, float wd) {
    g = g + wd * p;
    m = m + omb1 * (g - m);
    v = beta2 * v + omb2 * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p = p - step_size * (m / denom);
}

// `coef` (capturable form, cs_adam_step_dev): the per-tensor (lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)) pairs adam_prep_kernel left in
// device memory for this launch; NULL: the two by-value arguments.
__global__ __launch_bounds__(256) void adam_multi_kernel(const CsAdamTensor* __restrict__ tensors, AdamGrads grads, const int2* __restrict__ chunks,
                                                         int t0, float step_size, float inv_sqrt_bc2, float omb1, float beta2, float omb2, float eps, float wd,
                                                         const float2* __restrict__ coef) {
    const int2 ch = chunks[blockIdx.x];                      // (tensor index, first element / kAdamChunk)
    const CsAdamTensor t = tensors[ch.x];
    if (coef) {
        const float2 c2 = coef[ch.x];
        step_size = c2.x;
        inv_sqrt_bc2 = c2.y;
    }
    const float* __restrict__ g = nullptr;
    {
        const int li = ch.x - t0;
        // (static indexing of a by-value table: a runtime index would copy the whole table to scratch)
        const float* const* gt = grads.g;
        g = gt[li];
    }
    const long long begin = (long long)ch.y * kAdamChunk;
    long long end = begin + kAdamChunk;
    if (end > t.n) end = t.n;
    float* __restrict__ p = t.p;
    float* __restrict__ m = t.m;
    float* __restrict__ v = t.v;
    const bool vec = ((((unsigned long long)p) | ((unsigned long long)m) | ((unsigned long long)v) | ((unsigned long long)g)) & 15ull) == 0ull;
    if (vec) {
        const long long end4 = begin + ((end - begin) & ~3ll);
        for (long long i = begin + 4ll * threadIdx.x; i < end4; i += 1024) {
            float4 pp = *reinterpret_cast<const float4*>(p + i), mm = *reinterpret_cast<const float4*>(m + i);
            float4 vv = *reinterpret_cast<const float4*>(v + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            adam1(pp.x, mm.x, vv.x, gg.x, step_size, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd);
            adam1(pp.y, mm.y, vv.y, gg.y, step_size, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd);
            adam1(pp.z, mm.z, vv.z, gg.z, step_size, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd);
            adam1(pp.w, mm.w, vv.w, gg.w, step_size, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        }
        for (long long i = end4 + threadIdx.x; i < end; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam1(pp, mm, vv, g[i], step_size, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    } else {
        for (long long i = begin + threadIdx.x; i < end; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam1(pp, mm, vv, g[i], step_size, inv_sqrt_bc2, omb1, beta2, omb2, eps, wd);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

// Capturable form: the step counts live in DEVICE memory (one fp32 scalar per tensor: torch.optim.Adam(capturable=True)'s state
// layout), so a launch captured into a HIP graph advances them at every replay.  One thread per tensor: t = ++step, the bias
// corrections in double (device pow), the learning rate from a device double when the caller keeps one (schedulers change it
// between replays of a captured step).
__global__ void adam_prep_kernel(float* const* __restrict__ steps, int t0, int n, const double* __restrict__ lr_dev, double lr, double beta1,
                                 double beta2, float2* __restrict__ coef) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float* sp = steps[t0 + i];
    const float t = *sp + 1.0f;
    *sp = t;
    const double l = lr_dev ? *lr_dev : lr;
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    coef[t0 + i] = make_float2((float)(l / bc1), (float)(1.0 / sqrt(bc2)));
}

}  // namespace

extern "C" int cs_adam_chunk_elems(void) { return kAdamChunk; }
extern "C" int cs_adam_max_tensors(void) { return kAdamMaxTensors; }

extern "C" int cs_adam_step(const CsAdamTensor* tensors_dev, const void* const* grads_host, int t0, int n_tensors, const int* chunks_dev,
                            int n_chunks, double lr, double beta1, double beta2, double eps, double weight_decay, double step, void* stream) {
    CS_CHECK_ARG(tensors_dev && grads_host && chunks_dev && t0 >= 0 && n_tensors >= 1 && n_tensors <= kAdamMaxTensors && n_chunks >= 1,
                 "adam_step: 1..320 tensors per call, device tables, a HOST array of gradient pointers");
    CS_CHECK_ARG(step >= 1.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "adam_step: step >= 1, betas in [0, 1)");
    // scalar arithmetic in double, like torch's fused kernel: only the results are rounded to fp32
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    AdamGrads gr{};
    for (int i = 0; i < n_tensors; ++i) {
        CS_CHECK_ARG(grads_host[i], "adam_step: NULL gradient");
        gr.g[i] = reinterpret_cast<const float*>(grads_host[i]);
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tensors_dev, gr,
                       reinterpret_cast<const int2*>(chunks_dev), t0, step_size, inv_sqrt_bc2, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                       (float)weight_decay, static_cast<const float2*>(nullptr));
    CS_LAUNCH_CHECK();
    return CS_OK;
}

extern "C" int cs_adam_step_dev(const CsAdamTensor* tensors_dev, const void* const* grads_host, int t0, int n_tensors, const int* chunks_dev,
                                int n_chunks, float* const* steps_dev, float* coef_dev, const double* lr_dev, double lr, double beta1, double beta2,
                                double eps, double weight_decay, void* stream) {
    CS_CHECK_ARG(tensors_dev && grads_host && chunks_dev && steps_dev && coef_dev && t0 >= 0 && n_tensors >= 1 && n_tensors <= kAdamMaxTensors &&
                     n_chunks >= 1,
                 "adam_step_dev: 1..320 tensors per call, device tables (tensors, chunks, step pointers, coefficient scratch), a HOST array "
                 "of gradient pointers");
    CS_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "adam_step_dev: betas in [0, 1)");
    AdamGrads gr{};
    for (int i = 0; i < n_tensors; ++i) {
        CS_CHECK_ARG(grads_host[i], "adam_step_dev: NULL gradient");
        gr.g[i] = reinterpret_cast<const float*>(grads_host[i]);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_prep_kernel, dim3((unsigned)((n_tensors + 63) / 64)), dim3(64), 0, st, steps_dev, t0, n_tensors, lr_dev, lr, beta1, beta2,
                       reinterpret_cast<float2*>(coef_dev));
    CS_LAUNCH_CHECK();
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)n_chunks), dim3(256), 0, st, tensors_dev, gr, reinterpret_cast<const int2*>(chunks_dev), t0,
                       0.0f, 0.0f, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                       reinterpret_cast<const float2*>(coef_dev));
    CS_LAUNCH_CHECK();
    return CS_OK;
}
