/* c_abi_two_rank.c -- the ROW-SHARDED step of the path from plain C: one process per GPU, no Python, no torch, no MPI.
 *
 *   gcc -O2 -Iinclude examples/c_abi_two_rank.c -o /tmp/two_rank -Lhetmogp_amd -lhetmogp_hip -Wl,-rpath,$PWD/hetmogp_amd -lm
 *   for r in 0 1; do /tmp/two_rank $r 2 /tmp/hm_id.bin & done; wait          (a box with >= 2 MI355X)
 *
 * Rank 0 draws the ncclUniqueId through the library (hmogp_comm_unique_id) and publishes the 128 bytes through a FILE (write to
 * <idfile>.tmp, rename): any launcher that can hand 128 bytes from one process to the others will do.  Every rank uploads ONLY
 * its contiguous share of the rows of every task, attaches the communicator (hmogp_comm_init, device = rank) and
 * calls hmogp_elbo_grad_sharded: row pass on its rows -> wire pack -> ncclAllReduce (sum, fp64) -> unpack -> replicated finish,
 * all on the engine's own stream.  Rank 0 also evaluates ALL rows on a second, communicator-less engine and prints both results
 * (tests/test_dist_gpu.py compares them: every rank must report the same ELBO / gradients as the unsharded evaluation). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "hetmogp_hip.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != HMOGP_OK) {                                                                       \
      fprintf(stderr, "rank %d: %s failed: %d (%s)\n", rank, #call, rc_, hmogp_last_error(h));   \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

enum { T = 2, Q = 2, M = 128, P = 1, Df = 2, N0 = 6000, N1 = 5000, MTRI = M * (M + 1) / 2 };

static void shard(long long n, int rank, int world, long long* b, long long* e) { /* hetmogp_amd/dist.py:shard_rows */
  const long long base = n / world, extra = n % world;
  *b = rank * base + (rank < extra ? rank : extra);
  *e = *b + base + (rank < extra ? 1 : 0);
}

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <rank> <nranks> <idfile>\n", argv[0]);
    return 2;
  }
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  const char* idfile = argv[3];
  hmogp_handle h = NULL;
  const int32_t lik_id[T] = {HMOGP_LIK_GAUSSIAN, HMOGP_LIK_POISSON};
  const double lik_param[T] = {0.5, 0.0};
  const int32_t f_index[Df] = {0, 1}, d_index[Df] = {0, 0};
  hmogp_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = HMOGP_ABI_VERSION;
  cfg.T = T, cfg.Q = Q, cfg.M = M, cfg.P = P, cfg.Df = Df;
  cfg.lik_id = lik_id, cfg.lik_param = lik_param, cfg.f_index = f_index, cfg.d_index = d_index;
  cfg.device = rank, cfg.quirks = HMOGP_QUIRKS_REFERENCE;
  CHECK(hmogp_create(&cfg, &h));

  static double X0[N0], Y0[N0], X1[N1], Y1[N1];
  for (int i = 0; i < N0; ++i) X0[i] = (i + 0.5) / N0, Y0[i] = sin(7.0 * X0[i]) + 0.25 * cos(31.0 * i);
  for (int i = 0; i < N1; ++i) X1[i] = (i + 0.25) / N1, Y1[i] = floor(3.0 + 2.5 * sin(9.0 * X1[i]) + 1.4 * cos(13.0 * i));
  long long b0, e0, b1, e1;
  shard(N0, rank, world, &b0, &e0);
  shard(N1, rank, world, &b1, &e1);
  CHECK(hmogp_set_task_data(h, 0, X0 + b0, Y0 + b0, e0 - b0));          /* a rank uploads only its rows */
  CHECK(hmogp_set_task_data(h, 1, X1 + b1, Y1 + b1, e1 - b1));

  static double Z[M * Q * P], m_u[M * Q], L_flat[MTRI * Q], g_m_u[M * Q], g_L_u[MTRI * Q], g_Z[M * Q * P];
  double variance[Q] = {0.5, 0.7}, lengthscale[Q] = {1.0 / (M - 1), 1.3 / (M - 1)}, W[Q * Df] = {0.9, -0.4, 0.3, 0.8};
  double kappa[Q * Df] = {0, 0, 0, 0};
  for (int m = 0; m < M; ++m)
    for (int q = 0; q < Q; ++q) Z[m * Q + q] = m / (double)(M - 1), m_u[m * Q + q] = 0.5 * sin(1.0 + 3.0 * m + q);
  for (int r = 0, k = 0; r < M; ++r)
    for (int c = 0; c <= r; ++c, ++k)
      for (int q = 0; q < Q; ++q) L_flat[k * Q + q] = (r == c) ? 1.0 : 0.02 * cos(1.0 + r + 2.0 * c + q) / sqrt((double)M);
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

  /* the 128-byte id: rank 0 -> file -> everyone */
  char id[HMOGP_COMM_ID_BYTES];
  if (rank == 0) {
    char tmp[1024];
    CHECK(hmogp_comm_unique_id(id));
    snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 3;
    fclose(f);
    if (rename(tmp, idfile) != 0) return 3;
  } else {
    FILE* f = NULL;
    for (int tries = 0; tries < 6000 && !(f = fopen(idfile, "rb")); ++tries) usleep(10000);   /* up to 60 s */
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) {
      fprintf(stderr, "rank %d: no id file\n", rank);
      return 3;
    }
    fclose(f);
  }
  CHECK(hmogp_comm_init(h, world, rank, id));
  for (int it = 0; it < 3; ++it) CHECK(hmogp_elbo_grad_sharded(h, &prm, &out));
  double ms[HMOGP_NTIMINGS];
  CHECK(hmogp_last_timings(h, ms, NULL));
  double s_mu = 0.0, s_L = 0.0, s_Z = 0.0;
  for (int i = 0; i < M * Q; ++i) s_mu += g_m_u[i] * (1.0 + 0.001 * i);
  for (int i = 0; i < MTRI * Q; ++i) s_L += g_L_u[i] * (1.0 + 0.0001 * (i % 97));
  for (int i = 0; i < M * Q * P; ++i) s_Z += g_Z[i] * (1.0 + 0.001 * i);
  printf("rank %d sharded elbo %.17g g_var %.17g %.17g g_ell %.17g %.17g sums %.17g %.17g %.17g exchange_ms %.3f\n", rank, elbo,
         g_var[0], g_var[1], g_ell[0], g_ell[1], s_mu, s_L, s_Z, ms[8]);
  CHECK(hmogp_comm_destroy(h));
  hmogp_destroy(h);
  h = NULL;

  if (rank == 0) { /* the unsharded evaluation of ALL rows on one GPU: what every rank's result has to equal */
    CHECK(hmogp_create(&cfg, &h));
    CHECK(hmogp_set_task_data(h, 0, X0, Y0, N0));
    CHECK(hmogp_set_task_data(h, 1, X1, Y1, N1));
    CHECK(hmogp_elbo_grad(h, &prm, &out));
    s_mu = s_L = s_Z = 0.0;
    for (int i = 0; i < M * Q; ++i) s_mu += g_m_u[i] * (1.0 + 0.001 * i);
    for (int i = 0; i < MTRI * Q; ++i) s_L += g_L_u[i] * (1.0 + 0.0001 * (i % 97));
    for (int i = 0; i < M * Q * P; ++i) s_Z += g_Z[i] * (1.0 + 0.001 * i);
    printf("rank 0 single elbo %.17g g_var %.17g %.17g g_ell %.17g %.17g sums %.17g %.17g %.17g exchange_ms %.3f\n", elbo, g_var[0],
           g_var[1], g_ell[0], g_ell[1], s_mu, s_L, s_Z, 0.0);
    hmogp_destroy(h);
  }
  return 0;
}
