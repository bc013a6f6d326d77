// rocblas_sgemm accuracy probe for the three call shapes of pnrf_train.hip (diagnostic only)
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
static double relerr(const std::vector<float>& got, const std::vector<double>& want) {
  double a = 0, b = 0;
  for (size_t i = 0; i < got.size(); ++i) { a += (got[i] - want[i]) * (got[i] - want[i]); b += want[i] * want[i]; }
  return sqrt(a / b);
}
int main() {
  const int R = 1120, in = 256, out = 256;
  std::vector<float> X(R * in), W(out * in), dY(R * out);
  srand(1);
  for (auto& v : X) v = (rand() / (float)RAND_MAX - 0.3f);
  for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (auto& v : dY) v = (rand() / (float)RAND_MAX - 0.5f);
  float *dX_, *dW_, *dYd, *dXo, *dWo, *dYo;
  hipMalloc(&dX_, X.size() * 4); hipMalloc(&dW_, W.size() * 4); hipMalloc(&dYd, dY.size() * 4);
  hipMalloc(&dXo, X.size() * 4); hipMalloc(&dWo, W.size() * 4); hipMalloc(&dYo, dY.size() * 4);
  hipMemcpy(dX_, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW_, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dYd, dY.data(), dY.size() * 4, hipMemcpyHostToDevice);
  rocblas_handle h; rocblas_create_handle(&h);
  const float one = 1.f, zero = 0.f;
  // fwd: Y = X W^T
  rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, out, R, in, &one, dW_, in, dX_, in, &zero, dYo, out);
  // dx: dX = dY W
  rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, in, R, out, &one, dW_, in, dYd, out, &zero, dXo, in);
  // dw: dW = dY^T X
  rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, in, out, R, &one, dX_, in, dYd, out, &zero, dWo, in);
  hipDeviceSynchronize();
  std::vector<float> Yg(R * out), dXg(R * in), dWg(out * in);
  hipMemcpy(Yg.data(), dYo, Yg.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(dXg.data(), dXo, dXg.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(dWg.data(), dWo, dWg.size() * 4, hipMemcpyDeviceToHost);
  std::vector<double> Yw(R * out), dXw(R * in), dWw(out * in, 0.0);
  for (int r = 0; r < R; ++r) for (int o = 0; o < out; ++o) { double s = 0; for (int k = 0; k < in; ++k) s += (double)X[r * in + k] * W[o * in + k]; Yw[r * out + o] = s; }
  for (int r = 0; r < R; ++r) for (int k = 0; k < in; ++k) { double s = 0; for (int o = 0; o < out; ++o) s += (double)dY[r * out + o] * W[o * in + k]; dXw[r * in + k] = s; }
  for (int o = 0; o < out; ++o) for (int k = 0; k < in; ++k) { double s = 0; for (int r = 0; r < R; ++r) s += (double)dY[r * out + o] * X[r * in + k]; dWw[o * in + k] = s; }
  printf("fwd (T,N) rel err %.3e\ndx  (N,N) rel err %.3e\ndw  (N,T) rel err %.3e\n", relerr(Yg, Yw), relerr(dXg, dXw), relerr(dWg, dWw));
  return 0;
}
