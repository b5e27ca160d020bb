/* mbd_run.c — the reference's run_diffusion (mbd/planners/mbd_planner.py:38-182) and its seed sweep
 * (mbd/scripts/run_mbd.py:17-39) from plain C through include/mbd_hip.h: no Python, no MJCF compiler — envs are created
 * by NAME from the models embedded in libmbd_hip.so.
 *
 *   gcc -O2 -I include examples/mbd_run.c -o mbd_run -L model-based-diffusion_amd/lib -lmbd_hip -Wl,-rpath,$PWD/model-based-diffusion_amd/lib -lm
 *   ./mbd_run humanoidrun 1024 50 100 0.1 [n_seeds]
 *
 * Prints one line per seed: "seed S rew_final R steps_per_sec X", then (n_seeds > 1) the same seeds as ONE sweep. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mbd_hip.h"

#define CHECK(call)                                                                    \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if (rc_ != MBD_OK) {                                                               \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mbd_last_error());           \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

/* rng = PRNGKey(seed) (:40); rng, rng_reset = split(rng) (:79); rng_exp, rng = split(rng) (:150) */
static int key_chain(uint64_t seed, int impl, uint32_t rng_reset[2], uint32_t rng_exp[2]) {
  uint32_t key[2], k4[4];
  if (mbd_prng_key(seed, key) != MBD_OK) return 1;
  if (mbd_prng_split(key, 2, impl, k4) != MBD_OK) return 1;
  rng_reset[0] = k4[2]; rng_reset[1] = k4[3];
  const uint32_t rng[2] = {k4[0], k4[1]};
  if (mbd_prng_split(rng, 2, impl, k4) != MBD_OK) return 1;
  rng_exp[0] = k4[0]; rng_exp[1] = k4[1];
  return 0;
}

int main(int argc, char** argv) {
  const char* env_name = argc > 1 ? argv[1] : "humanoidrun";
  const int N = argc > 2 ? atoi(argv[2]) : 1024, H = argc > 3 ? atoi(argv[3]) : 50, Nd = argc > 4 ? atoi(argv[4]) : 100;
  const float temp = argc > 5 ? (float)atof(argv[5]) : 0.1f;
  const int n_seeds = argc > 6 ? atoi(argv[6]) : 1;
  const int impl = MBD_PRNG_PARTITIONABLE;
  mbd_env* env = NULL;
  CHECK(mbd_env_create(env_name, 0, &env));
  int Nu = 0, Nx = 0, S = 0;
  CHECK(mbd_env_info(env, &Nu, &Nx, &S, NULL, NULL, NULL));
  mbd_plan_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.Nsample = N; cfg.Hsample = H; cfg.Ndiffuse = Nd; cfg.temp_sample = temp;
  cfg.beta0 = 1e-4f; cfg.betaT = 1e-2f; cfg.prng_impl = impl; cfg.shard_begin = 0; cfg.shard_count = N;
  cfg.literal_score = 1;
  float* state = (float*)malloc(sizeof(float) * (size_t)S * (size_t)(n_seeds > 0 ? n_seeds : 1));
  uint32_t* keys = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)(n_seeds > 0 ? n_seeds : 1));
  for (int seed = 0; seed < n_seeds; ++seed) {
    uint32_t rng_reset[2], rng_exp[2];
    if (key_chain((uint64_t)seed, impl, rng_reset, rng_exp)) return 1;
    CHECK(mbd_env_reset(env, rng_reset, impl, state + (size_t)seed * S));
    keys[2 * seed] = rng_exp[0]; keys[2 * seed + 1] = rng_exp[1];
    mbd_plan* plan = NULL;
    CHECK(mbd_plan_create(env, &cfg, &plan));
    CHECK(mbd_plan_set_state0(plan, state + (size_t)seed * S));
    float rew_final = 0.0f;
    double secs = 0.0;
    CHECK(mbd_plan_run(plan, rng_exp, NULL, NULL, &rew_final, &secs));
    printf("seed %d rew_final %.9g steps_per_sec %.1f\n", seed, rew_final, (Nd - 1) / secs);
    CHECK(mbd_plan_destroy(plan));
  }
  if (n_seeds > 1) { /* the same plans in lockstep: one rollout launch per diffusion step over all candidates */
    mbd_sweep* sweep = NULL;
    CHECK(mbd_sweep_create(env, &cfg, n_seeds, NULL, &sweep));
    for (int seed = 0; seed < n_seeds; ++seed) CHECK(mbd_sweep_set_state0(sweep, seed, state + (size_t)seed * S));
    float* rews = (float*)malloc(sizeof(float) * (size_t)n_seeds);
    double secs = 0.0;
    CHECK(mbd_sweep_run(sweep, keys, NULL, NULL, rews, &secs));
    for (int seed = 0; seed < n_seeds; ++seed) printf("sweep seed %d rew_final %.9g\n", seed, rews[seed]);
    printf("sweep plan_steps_per_sec %.1f\n", (double)n_seeds * (Nd - 1) / secs);
    free(rews);
    CHECK(mbd_sweep_destroy(sweep));
  }
  free(keys);
  free(state);
  CHECK(mbd_env_destroy(env));
  return 0;
}
