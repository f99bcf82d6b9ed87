/* A forecast roll-out from a host program WITHOUT Python: plain C99 over include/aurora_hip.h and the HIP runtime's C API.
 *
 * What the reference does in `rollout(model, batch, steps)` (aurora/rollout.py:14-49) around `Aurora.forward`
 * (aurora/model/aurora.py:265-392) happens here around aurora_hip_step: crop the surplus latitude row (batch.py:142-168),
 * step, drop the oldest history state, append the prediction, advance the time stamp.  Weights come from a packed weight
 * file (aurora_hip_save_packed; no pickle, no checkpoint adapters on this side).
 *
 *   aurora_forecast <case-dir>
 *   aurora_forecast <case-dir> --rank R --world N --nccl-id FILE [--run-nonce N] [--device D]      (built with -DAURORA_WITH_RCCL;
 *                   N: the same non-zero number for every rank of ONE run -- a stale bootstrap file is then never taken for it)
 *
 * The second form runs ONE latitude band of the forecast: N processes, one per GPU, started by any
 * launcher (a shell loop will do); rank 0 writes RCCL's unique id to FILE, the others read it; the halo rows of the
 * shifted-window blocks travel over RCCL point-to-point (examples/c_host/rccl_transport.c).  Every rank reads its own rows
 * of the input files and writes pred<step>_..._rank<R>.f32 with its rows of the prediction.
 *
 * <case-dir>/case.txt     whitespace-separated `key value...` records (see read_case below): the Aurora.__init__ keywords
 *                          of the ERA5 model family, the grid, the normalisation statistics, B, T, steps, time stamps
 * <case-dir>/weights.aurorahip
 * <case-dir>/surf_<var>.f32   (B, T, H, W)      little-endian float32, C order
 * <case-dir>/static_<var>.f32 (H, W)
 * <case-dir>/atmos_<var>.f32  (B, T, C, H, W)
 * writes <case-dir>/pred<step>_surf_<var>.f32 (B, H', W) and pred<step>_atmos_<var>.f32 (B, C, H', W).
 *
 * Build (tests/test_c_host.py does exactly this):
 *   gcc -std=c99 -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/aurora_forecast.c \
 *       -o aurora_forecast -Laurora_amd/_lib -laurora_hip -L/opt/rocm/lib -lamdhip64 -lm
 *   (band mode: add -DAURORA_WITH_RCCL -D_POSIX_C_SOURCE=200809L examples/c_host/rccl_transport.c -lrccl)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aurora_hip.h"
#ifdef AURORA_WITH_RCCL
#include "rccl_transport.h"
#endif

#define MAX_VARS 32
#define MAX_DIM 8192

static void die(const char* what, const char* detail) {
  fprintf(stderr, "aurora_forecast: %s: %s\n", what, detail ? detail : "");
  exit(1);
}
#define HIP(call)                                                        \
  do {                                                                   \
    hipError_t e_ = (call);                                              \
    if (e_ != hipSuccess) die(#call, hipGetErrorString(e_));             \
  } while (0)
#define AUR(call)                                                        \
  do {                                                                   \
    if ((call) != 0) die(#call, aurora_hip_last_error());                \
  } while (0)

typedef struct names { int n; char* v[MAX_VARS]; } names;

typedef struct forecast_case {
  aurora_hip_config cfg;
  names surf, stat, atmos;
  int n_lat, n_lon, n_levels, levels_float32, B, T, steps;
  double lat[MAX_DIM], lon[MAX_DIM], levels[64], time_hours[64];
  double surf_loc[MAX_VARS], surf_scale[MAX_VARS], static_loc[MAX_VARS], static_scale[MAX_VARS];
  double atmos_loc[MAX_VARS * 64], atmos_scale[MAX_VARS * 64];
} forecast_case;

static int read_ints(FILE* f, int32_t* out, int cap) {
  int n = 0;
  if (fscanf(f, "%d", &n) != 1 || n < 0 || n > cap) die("case.txt", "bad integer list");
  for (int i = 0; i < n; ++i)
    if (fscanf(f, "%d", &out[i]) != 1) die("case.txt", "bad integer");
  return n;
}
static int read_doubles(FILE* f, double* out, int cap) {
  int n = 0;
  if (fscanf(f, "%d", &n) != 1 || n < 0 || n > cap) die("case.txt", "bad number list");
  for (int i = 0; i < n; ++i)
    if (fscanf(f, "%lf", &out[i]) != 1) die("case.txt", "bad number");
  return n;
}
static void read_names(FILE* f, names* out) {
  if (fscanf(f, "%d", &out->n) != 1 || out->n < 0 || out->n > MAX_VARS) die("case.txt", "bad name list");
  for (int i = 0; i < out->n; ++i) {
    char buf[128];
    if (fscanf(f, "%127s", buf) != 1) die("case.txt", "bad name");
    out->v[i] = (char*)malloc(strlen(buf) + 1);
    strcpy(out->v[i], buf);
  }
}

/* Records: scalars `key value`; lists `key n v0 v1 ...`. */
static void read_case(const char* dir, forecast_case* c) {
  char path[4096], key[128];
  snprintf(path, sizeof path, "%s/case.txt", dir);
  FILE* f = fopen(path, "r");
  if (!f) die("cannot open", path);
  memset(c, 0, sizeof *c);
  aurora_hip_config* g = &c->cfg;
  while (fscanf(f, "%127s", key) == 1) {
#define SCALAR(name, fmt, dst) else if (!strcmp(key, name)) { if (fscanf(f, fmt, dst) != 1) die("case.txt", name); }
    if (0) {}
    SCALAR("embed_dim", "%d", &g->embed_dim) SCALAR("patch_size", "%d", &g->patch_size)
    SCALAR("latent_levels", "%d", &g->latent_levels) SCALAR("num_heads", "%d", &g->num_heads)
    SCALAR("enc_depth", "%d", &g->enc_depth) SCALAR("dec_depth", "%d", &g->dec_depth)
    SCALAR("perceiver_ln_eps", "%f", &g->perceiver_ln_eps) SCALAR("max_history", "%d", &g->max_history)
    SCALAR("timestep_hours", "%lf", &g->timestep_hours) SCALAR("stabilise_level_agg", "%d", &g->stabilise_level_agg)
    SCALAR("use_lora", "%d", &g->use_lora) SCALAR("lora_steps", "%d", &g->lora_steps)
    SCALAR("lora_mode", "%d", &g->lora_mode) SCALAR("autocast", "%d", &g->autocast)
    SCALAR("levels_float32", "%d", &c->levels_float32) SCALAR("B", "%d", &c->B) SCALAR("T", "%d", &c->T)
    SCALAR("steps", "%d", &c->steps)
#undef SCALAR
    else if (!strcmp(key, "encoder_depths")) g->n_stages = read_ints(f, g->encoder_depths, 4);
    else if (!strcmp(key, "encoder_heads")) read_ints(f, g->encoder_heads, 4);
    else if (!strcmp(key, "decoder_depths")) read_ints(f, g->decoder_depths, 4);
    else if (!strcmp(key, "decoder_heads")) read_ints(f, g->decoder_heads, 4);
    else if (!strcmp(key, "window")) read_ints(f, g->window, 3);
    else if (!strcmp(key, "surf_vars")) read_names(f, &c->surf);
    else if (!strcmp(key, "static_vars")) read_names(f, &c->stat);
    else if (!strcmp(key, "atmos_vars")) read_names(f, &c->atmos);
    else if (!strcmp(key, "lat")) c->n_lat = read_doubles(f, c->lat, MAX_DIM);
    else if (!strcmp(key, "lon")) c->n_lon = read_doubles(f, c->lon, MAX_DIM);
    else if (!strcmp(key, "levels")) c->n_levels = read_doubles(f, c->levels, 64);
    else if (!strcmp(key, "time_hours")) read_doubles(f, c->time_hours, 64);
    else if (!strcmp(key, "surf_loc")) read_doubles(f, c->surf_loc, MAX_VARS);
    else if (!strcmp(key, "surf_scale")) read_doubles(f, c->surf_scale, MAX_VARS);
    else if (!strcmp(key, "static_loc")) read_doubles(f, c->static_loc, MAX_VARS);
    else if (!strcmp(key, "static_scale")) read_doubles(f, c->static_scale, MAX_VARS);
    else if (!strcmp(key, "atmos_loc")) read_doubles(f, c->atmos_loc, MAX_VARS * 64);
    else if (!strcmp(key, "atmos_scale")) read_doubles(f, c->atmos_scale, MAX_VARS * 64);
    else die("case.txt: unknown key", key);
  }
  fclose(f);
  g->n_surf = c->surf.n, g->n_static = c->stat.n, g->n_atmos = c->atmos.n;
  g->surf_vars = (const char* const*)c->surf.v;
  g->static_vars = (const char* const*)c->stat.v;
  g->atmos_vars = (const char* const*)c->atmos.v;
}

/* Reads `planes` images of (n_lat, n_lon) and uploads their latitude rows [row0, row0 + rows), packed. */
static float* upload_rows(const char* dir, const char* kind, const char* var, int64_t planes, int n_lat, int row0, int rows, int n_lon) {
  char path[4096];
  snprintf(path, sizeof path, "%s/%s_%s.f32", dir, kind, var);
  FILE* f = fopen(path, "rb");
  if (!f) die("cannot open", path);
  const size_t image = (size_t)n_lat * n_lon, kept = (size_t)rows * n_lon;
  float* host = (float*)malloc(sizeof(float) * image);
  float* dev = NULL;
  HIP(hipMalloc((void**)&dev, sizeof(float) * kept * planes));
  for (int64_t p = 0; p < planes; ++p) {
    if (fread(host, sizeof(float), image, f) != image) die("short file", path);
    HIP(hipMemcpy(dev + p * kept, host + (size_t)row0 * n_lon, sizeof(float) * kept, hipMemcpyHostToDevice));
  }
  free(host);
  fclose(f);
  return dev;
}

static void download(const char* dir, int step, const char* kind, const char* var, int rank, const float* dev, size_t n) {
  char path[4096];
  if (rank < 0) snprintf(path, sizeof path, "%s/pred%d_%s_%s.f32", dir, step, kind, var);
  else snprintf(path, sizeof path, "%s/pred%d_%s_%s_rank%d.f32", dir, step, kind, var, rank);
  float* host = (float*)malloc(sizeof(float) * n);
  HIP(hipMemcpy(host, dev, sizeof(float) * n, hipMemcpyDeviceToHost));
  FILE* f = fopen(path, "wb");
  if (!f || fwrite(host, sizeof(float), n, f) != n) die("cannot write", path);
  fclose(f);
  free(host);
}

/* History update of rollout.py:39-49 on the device: state t <- state t + 1, the newest <- the prediction.
 * hist: (B, T, plane), pred: (B, plane); the copies are ordered on `stream` behind the step that wrote `pred`. */
static void push_history(float* hist, const float* pred, int B, int T, size_t plane, hipStream_t stream) {
  for (int b = 0; b < B; ++b) {
    float* h = hist + (size_t)b * T * plane;
    for (int t = 0; t + 1 < T; ++t)
      HIP(hipMemcpyAsync(h + (size_t)t * plane, h + (size_t)(t + 1) * plane, sizeof(float) * plane, hipMemcpyDeviceToDevice, stream));
    HIP(hipMemcpyAsync(h + (size_t)(T - 1) * plane, pred + (size_t)b * plane, sizeof(float) * plane, hipMemcpyDeviceToDevice, stream));
  }
}

int main(int argc, char** argv) {
  int rank = -1, world = 1, device = -1;
  const char* id_file = NULL;
  unsigned long long run_nonce = 0;
  int bad = argc < 2;
  for (int i = 2; i < argc && !bad; i += 2) {
    if (i + 1 >= argc) bad = 1;
    else if (!strcmp(argv[i], "--rank")) rank = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--world")) world = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--nccl-id")) id_file = argv[i + 1];
    else if (!strcmp(argv[i], "--run-nonce")) run_nonce = strtoull(argv[i + 1], NULL, 10);
    else bad = 1;
  }
  if (world > 1 && (rank < 0 || rank >= world || !id_file)) bad = 1;
  if (bad) {
    fprintf(stderr, "usage: aurora_forecast <case-dir> [--rank R --world N --nccl-id FILE [--run-nonce N] [--device D]]   (library ABI version %d)\n",
            aurora_hip_version());
    return 2;
  }
#ifndef AURORA_WITH_RCCL
  (void)run_nonce;
  if (world > 1) die("band mode", "this binary was built without -DAURORA_WITH_RCCL");
#endif
  if (world <= 1) rank = -1;
  const char* dir = argv[1];
  static forecast_case c;
  read_case(dir, &c);
  int32_t sizes[16];
  if (aurora_hip_abi_sizes(sizes, 16) < 3 || sizes[0] != (int32_t)sizeof(aurora_hip_config) ||
      sizes[1] != (int32_t)sizeof(aurora_hip_grid) || sizes[2] != (int32_t)sizeof(aurora_hip_step_io))
    die("ABI", "struct sizes of this build differ from the library's");

  HIP(hipSetDevice(device >= 0 ? device : (rank > 0 ? rank : 0)));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));

  aurora_hip_model* model = NULL;
  char path[4096];
  snprintf(path, sizeof path, "%s/weights.aurorahip", dir);
  AUR(aurora_hip_create(&c.cfg, &model));
  AUR(aurora_hip_load_packed(model, path));
  AUR(aurora_hip_finalize(model, stream));
#ifdef AURORA_WITH_RCCL
  static rccl_transport transport;
  if (world > 1) {   /* call order of include/aurora_hip.h: set_band, precompute with the FULL grid, band_rows, staging */
    if (rccl_transport_init(&transport, rank, world, id_file, run_nonce, 120.0) != 0) die("RCCL", transport.error);
    aurora_hip_band band;
    band.rank = rank, band.world = world;
    band.post = rccl_transport_post, band.wait = rccl_transport_wait, band.user = &transport;
    AUR(aurora_hip_set_band(model, &band));
  }
#endif

  /* Batch.crop: a grid with one latitude row more than a multiple of the patch size loses its last row */
  const int P = c.cfg.patch_size, H = c.n_lat - (c.n_lat % P == 1 ? 1 : 0), W = c.n_lon, C = c.n_levels, B = c.B, T = c.T;
  aurora_hip_grid grid;
  memset(&grid, 0, sizeof grid);
  grid.n_lat = H, grid.n_lon = W, grid.lat = c.lat, grid.lon = c.lon;
  grid.n_levels = C, grid.levels = c.levels, grid.levels_float32 = c.levels_float32;
  grid.surf_loc = c.surf_loc, grid.surf_scale = c.surf_scale;
  grid.static_loc = c.static_loc, grid.static_scale = c.static_scale;
  grid.atmos_loc = c.atmos_loc, grid.atmos_scale = c.atmos_scale;
  AUR(aurora_hip_precompute(model, &grid, stream));
  int row0 = 0, Hb = H;   /* this process's latitude rows: the whole (cropped) grid, or its band */
#ifdef AURORA_WITH_RCCL
  if (world > 1) {
    int32_t r0 = 0, r1 = 0;
    AUR(aurora_hip_band_rows(model, &r0, &r1));
    row0 = r0, Hb = r1 - r0;
    if (rccl_transport_allocate(&transport, aurora_hip_band_staging_bytes(model)) != 0) die("RCCL", transport.error);
    AUR(aurora_hip_set_band_staging(model, transport.send, transport.recv, transport.staging_bytes));
    if (rccl_transport_selftest(&transport, 1 << 20, stream) != 0) die("RCCL self-test", transport.error);
    fprintf(stderr, "aurora_forecast: rank %d of %d owns latitude rows [%d, %d), halo staging %.1f MiB, transport self-test ok\n",
            rank, world, row0, row0 + Hb, (double)transport.staging_bytes / 1048576.0);
  }
#endif

  const size_t plane = (size_t)Hb * W;
  float *surf[MAX_VARS], *stat[MAX_VARS], *atmos[MAX_VARS], *out_surf[MAX_VARS], *out_atmos[MAX_VARS];
  for (int i = 0; i < c.surf.n; ++i) {
    surf[i] = upload_rows(dir, "surf", c.surf.v[i], (int64_t)B * T, c.n_lat, row0, Hb, W);
    HIP(hipMalloc((void**)&out_surf[i], sizeof(float) * B * plane));
  }
  for (int i = 0; i < c.stat.n; ++i) stat[i] = upload_rows(dir, "static", c.stat.v[i], 1, c.n_lat, row0, Hb, W);
  for (int i = 0; i < c.atmos.n; ++i) {
    atmos[i] = upload_rows(dir, "atmos", c.atmos.v[i], (int64_t)B * T * C, c.n_lat, row0, Hb, W);
    HIP(hipMalloc((void**)&out_atmos[i], sizeof(float) * B * C * plane));
  }

  aurora_hip_step_io io;
  memset(&io, 0, sizeof io);
  io.B = B, io.T = T;
  io.surf = (const float* const*)surf, io.stat = (const float* const*)stat, io.atmos = (const float* const*)atmos;
  io.out_surf = out_surf, io.out_atmos = out_atmos;
  io.surf_strides[0] = (int64_t)T * plane, io.surf_strides[1] = plane, io.surf_strides[2] = W, io.surf_strides[3] = 1;
  io.static_strides[0] = W, io.static_strides[1] = 1;
  io.atmos_strides[0] = (int64_t)T * C * plane, io.atmos_strides[1] = (int64_t)C * plane, io.atmos_strides[2] = plane;
  io.atmos_strides[3] = W, io.atmos_strides[4] = 1;

  double hours[64];
  memcpy(hours, c.time_hours, sizeof hours);
  for (int step = 0; step < c.steps; ++step) {
    io.rollout_step = step;
    AUR(aurora_hip_set_time(model, hours, B, stream));
    AUR(aurora_hip_step(model, &io, stream));
    HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < c.surf.n; ++i) download(dir, step, "surf", c.surf.v[i], rank, out_surf[i], (size_t)B * plane);
    for (int i = 0; i < c.atmos.n; ++i) download(dir, step, "atmos", c.atmos.v[i], rank, out_atmos[i], (size_t)B * C * plane);
    for (int i = 0; i < c.surf.n; ++i) push_history(surf[i], out_surf[i], B, T, plane, stream);
    for (int i = 0; i < c.atmos.n; ++i) push_history(atmos[i], out_atmos[i], B, T, (size_t)C * plane, stream);
    for (int b = 0; b < B; ++b) hours[b] += c.cfg.timestep_hours;
  }
  HIP(hipStreamSynchronize(stream));
  printf("aurora_forecast: %d step(s) of a %d x %d x %d grid, workspace %.1f MiB\n", c.steps, C, Hb, W,
         (double)aurora_hip_workspace_bytes(model) / 1048576.0);
#ifdef AURORA_WITH_RCCL
  if (world > 1) {
    printf("aurora_forecast: rank %d: %lld halo exchanges, %.1f MiB sent\n", rank, (long long)transport.exchanges,
           (double)transport.bytes_sent / 1048576.0);
    aurora_hip_set_band(model, NULL);
    rccl_transport_destroy(&transport);
  }
#endif
  aurora_hip_destroy(model);
  HIP(hipStreamDestroy(stream));
  return 0;
}
