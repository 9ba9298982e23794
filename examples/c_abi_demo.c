/* c_abi_demo.c -- the drop-in boundary used from plain C: no Python, no torch, only include/hetmogp_hip.h.
 *
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -o /tmp/c_abi_demo -Lhetmogp_amd -lhetmogp_hip -Wl,-rpath,$PWD/hetmogp_amd -lm
 *   /tmp/c_abi_demo            (on an MI355X)
 *
 * T = 2 [Gaussian(sigma 0.5), Bernoulli], N = 400 / 300 rows on a grid, M = 16 inducing points, Q = 2 latent GPs: one
 * hmogp_elbo_grad (= SVMOGP.parameters_changed(), hetmogp/svmogp.py:85-166), then the same evaluation as a "row-sharded run
 * of one rank" with the exchange step inside the library (hmogp_comm_*), which must give the same bits.  Prints the ELBO and
 * a few gradient entries with 17 significant digits (tests/test_demo_gpu.py compares them with the ctypes binding's). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hetmogp_hip.h"

#define CHECK(call)                                                                 \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != HMOGP_OK) {                                                          \
      fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, hmogp_last_error(h));     \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main(void) {
  enum { T = 2, Q = 2, M = 16, P = 1, Df = 2, N0 = 400, N1 = 300, MTRI = M * (M + 1) / 2 };
  const int32_t lik_id[T] = {HMOGP_LIK_GAUSSIAN, HMOGP_LIK_BERNOULLI};
  const double lik_param[T] = {0.5, 0.0};
  const int32_t f_index[Df] = {0, 1}, d_index[Df] = {0, 0};
  hmogp_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = HMOGP_ABI_VERSION;
  cfg.T = T, cfg.Q = Q, cfg.M = M, cfg.P = P, cfg.Df = Df;
  cfg.lik_id = lik_id, cfg.lik_param = lik_param, cfg.f_index = f_index, cfg.d_index = d_index;
  cfg.device = 0, cfg.chunk_rows = 0, cfg.quirks = HMOGP_QUIRKS_REFERENCE;
  /* regular kernels also for this small model: a sharded step (hmogp_elbo_grad_sharded) never takes the fused small-model
   * kernels, and the demo shows that its one-rank exchange is BIT-identical to the plain call */
  cfg.flags = HMOGP_CFG_NO_SMALL_PATH;
  hmogp_handle h = NULL;
  CHECK(hmogp_create(&cfg, &h));

  /* deterministic data (the test rebuilds exactly these arrays in NumPy) */
  static double X0[N0], Y0[N0], X1[N1], Y1[N1];
  for (int i = 0; i < N0; ++i) X0[i] = (i + 0.5) / N0, Y0[i] = sin(7.0 * X0[i]) + 0.25 * cos(31.0 * i);
  for (int i = 0; i < N1; ++i) X1[i] = (i + 0.25) / N1, Y1[i] = (sin(5.0 * X1[i]) + 0.3 * cos(17.0 * i) > 0.0) ? 1.0 : 0.0;
  CHECK(hmogp_set_task_data(h, 0, X0, Y0, N0));
  CHECK(hmogp_set_task_data(h, 1, X1, Y1, N1));

  static double Z[M * Q * P], m_u[M * Q], L_flat[MTRI * Q], g_m_u[M * Q], g_L_u[MTRI * Q], g_Z[M * Q * P];
  double variance[Q] = {0.5, 0.7}, lengthscale[Q] = {0.08, 0.11}, W[Q * Df] = {0.9, -0.4, 0.3, 0.8}, kappa[Q * Df] = {0, 0, 0, 0};
  for (int m = 0; m < M; ++m)
    for (int q = 0; q < Q; ++q) {
      Z[m * Q + q] = m / (double)(M - 1);
      m_u[m * Q + q] = 0.5 * sin(1.0 + 3.0 * m + q);
    }
  for (int r = 0, k = 0; r < M; ++r)
    for (int c = 0; c <= r; ++c, ++k)
      for (int q = 0; q < Q; ++q) L_flat[k * Q + q] = (r == c) ? 1.0 : 0.02 * cos(1.0 + r + 2.0 * c + q);

  hmogp_params prm;
  memset(&prm, 0, sizeof prm);
  prm.Z = Z, prm.m_u = m_u, prm.L_flat = L_flat, prm.variance = variance, prm.lengthscale = lengthscale, prm.W = W, prm.kappa = kappa;
  prm.group_mask = HMOGP_GROUP_ALL;
  double elbo = 0.0, g_var[Q], g_ell[Q], g_W[Q * Df], g_kap[Q * Df], kl[Q];
  int32_t rung[Q];
  uint32_t flags = 0;
  hmogp_outputs out;
  memset(&out, 0, sizeof out);
  out.elbo = &elbo, out.g_m_u = g_m_u, out.g_L_u = g_L_u, out.g_variance = g_var, out.g_lengthscale = g_ell, out.g_W = g_W;
  out.g_kappa = g_kap, out.g_Z = g_Z, out.rung = rung, out.flags = &flags, out.kl = kl;
  CHECK(hmogp_elbo_grad(h, &prm, &out));
  printf("elbo %.17g\n", elbo);
  printf("g_variance %.17g %.17g\n", g_var[0], g_var[1]);
  printf("g_lengthscale %.17g %.17g\n", g_ell[0], g_ell[1]);
  printf("g_m_u[0] %.17g g_L_u[5] %.17g g_Z[3] %.17g\n", g_m_u[0], g_L_u[5], g_Z[3]);

  /* the same step with the exchange inside the library: a communicator of ONE rank runs pack -> ncclAllReduce -> unpack */
  if (hmogp_comm_available()) {
    char id[HMOGP_COMM_ID_BYTES];
    const double elbo_plain = elbo, g5 = g_L_u[5];
    CHECK(hmogp_comm_unique_id(id));
    CHECK(hmogp_comm_init(h, 1, 0, id));
    CHECK(hmogp_elbo_grad_sharded(h, &prm, &out));
    double ms[HMOGP_NTIMINGS];
    CHECK(hmogp_last_timings(h, ms, NULL));
    printf("native exchange: identical %d, %.3f ms on the engine's stream\n", elbo == elbo_plain && g_L_u[5] == g5, ms[8]);
    CHECK(hmogp_comm_destroy(h));
  } else {
    printf("native exchange: librccl not loadable\n");
  }
  hmogp_destroy(h);
  return 0;
}
