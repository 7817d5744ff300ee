// Kaldi-compatible log-mel filterbank front end of the audio Lens on the GPU.
//
//   vl_kaldi_fbank  = torchaudio.compliance.kaldi.fbank(waveform, htk_compat=True, sample_frequency=16000, use_energy=False,
//                     window_type="hanning", num_mel_bins=128, dither=0.0, frame_shift=10) as called by
//                     AudioASTProcessorEval.convert2fbank (open_clip/modal_audio/processors/at_processor.py:854-873), followed by
//                     the zero-padding / truncation to target_length rows and transforms.Normalize(mean, std) of the same
//                     processor (:839-851, :864-871).  Kaldi's published algorithm (snip_edges framing, per-frame DC removal,
//                     pre-emphasis 0.97 with the first sample replicated, povey-free "hanning" window, zero-padding to the
//                     next power of two, power spectrum, triangular mel filters between 20 Hz and Nyquist on the
//                     1127 ln(1 + f/700) scale, natural log floored at FLT_EPSILON).
//
// One workgroup per frame: the 25 ms window goes through LDS, every thread owns one DFT bin (512-point direct DFT against
// a sin/cos table indexed by (k*n) mod 512 - exact phase reduction - accumulated in double: 0.1 MFLOP per frame, nothing to
// optimise), then one mel bin.  The window and the mel filter matrix come from the host (vitlens_hip/audio.py builds them
// as the published formulas say; oracle/fbank_oracle.py is the numpy restatement the tests compare with - torchaudio is
// not installed anywhere, so this path is "parity unpinned": restated from the algorithm, not checked against the library).
#include "vl_common.h"
#include "vitlens_hip.h"

namespace {

__global__ void __launch_bounds__(256) kaldi_fbank_kernel(const float* wave, long wave_stride, const float* window, const float* banks,
                                                          float* out, long out_stride, int n_frames, int target_len, int win, int shift,
                                                          int nfft, int nmel, float preemph, float mean, float inv_std) {
  extern __shared__ float s_fb[];
  float* y = s_fb;                       // [nfft] windowed frame, zero padded
  float* tc = y + nfft;                  // [nfft] cos(2 pi j / nfft)
  float* ts = tc + nfft;                 // [nfft] sin(2 pi j / nfft)
  float* pw = ts + nfft;                 // [nfft/2 + 1] power spectrum
  __shared__ float s_red[4];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  float* o = out + (size_t)b * out_stride + (size_t)f * nmel;
  if (f >= n_frames) {                   // padded rows: Normalize of a zero row
    for (int m = tid; m < nmel; m += 256) o[m] = (0.f - mean) * inv_std;
    return;
  }
  const float* x = wave + (size_t)b * wave_stride + (size_t)f * shift;
  // DC removal: the frame's mean
  float part = 0.f;
  for (int i = tid; i < win; i += 256) part += x[i];
  part = wave_sum(part);
  if ((tid & 63) == 0) s_red[tid >> 6] = part;
  for (int j = tid; j < nfft; j += 256) {
    float sv, cv;
    sincospif(2.0f * (float)j / (float)nfft, &sv, &cv);
    tc[j] = cv; ts[j] = sv;
  }
  __syncthreads();
  const float mu = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)win;
  for (int i = tid; i < nfft; i += 256) {
    float v = 0.f;
    if (i < win) {
      const float cur = x[i] - mu, prev = x[i > 0 ? i - 1 : 0] - mu;      // pre-emphasis with the first sample replicated
      v = (cur - preemph * prev) * window[i];
    }
    y[i] = v;
  }
  __syncthreads();
  const int nb = nfft / 2 + 1;
  for (int k = tid; k < nb; k += 256) {
    double re = 0.0, im = 0.0;
    for (int n = 0; n < win; ++n) {
      const int j = (k * n) & (nfft - 1);
      re += (double)y[n] * (double)tc[j];
      im -= (double)y[n] * (double)ts[j];
    }
    pw[k] = (float)(re * re + im * im);
  }
  __syncthreads();
  for (int m = tid; m < nmel; m += 256) {
    const float* w = banks + (size_t)m * nb;
    float e = 0.f;
    for (int k = 0; k < nb; ++k) e = fmaf(pw[k], w[k], e);
    o[m] = (logf(fmaxf(e, 1.1920928955078125e-07f)) - mean) * inv_std;
  }
}

}  // namespace

extern "C" int vl_set_error(const char* msg);

extern "C" int vl_kaldi_fbank(const float* wave, long wave_stride, int batch, long n_samples, const float* window, const float* banks,
                              float* out, int target_len, int win, int shift, int nfft, int nmel, float preemph, float mean,
                              float std, hipStream_t stream) {
  if (batch <= 0 || n_samples < win || win <= 0 || shift <= 0 || nmel <= 0 || target_len <= 0 || std == 0.f)
    return vl_set_error("vl_kaldi_fbank: bad shape (at least one full window of samples is needed)");
  if (nfft < win || (nfft & (nfft - 1)) || nfft > 4096) return vl_set_error("vl_kaldi_fbank: nfft must be a power of two in [win, 4096]");
  const int n_frames = (int)(1 + (n_samples - win) / shift);            // snip_edges = True
  const int rows = n_frames < target_len ? n_frames : target_len;
  const size_t smem = (size_t)(3 * nfft + nfft / 2 + 1) * sizeof(float);
  hipLaunchKernelGGL(kaldi_fbank_kernel, dim3(target_len, batch), dim3(256), smem, stream, wave, wave_stride, window, banks, out,
                     (long)target_len * nmel, rows, target_len, win, shift, nfft, nmel, preemph, mean, 1.0f / std);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : vl_set_error(hipGetErrorString(e));
}
