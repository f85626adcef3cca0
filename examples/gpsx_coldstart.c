/* examples/gpsx_coldstart.c -- a plain C host on the batched C ABI (include/gpsx.h): cold-start acquisition of a raw IF
 * recording.  All 32 PRNs x 21 Doppler bins (-5 .. +5 kHz) x 16368 code phases, non-coherently over n_ms consecutive
 * milliseconds, in ONE call -- the sweep the firmware spreads over minutes of acquisition_freq_search /
 * acquisition_code_phase_search steps (PM/GPS/acquisition.c:196-312) -- then the satellites whose peak stands out.
 *
 *   gcc -O2 -I include examples/gpsx_coldstart.c -L stm32f4_sdr_gps_amd/lib -lgpsx \
 *       -Wl,-rpath,$PWD/stm32f4_sdr_gps_amd/lib -o gpsx_coldstart
 *   ./gpsx_coldstart capture.bin [n_ms = 4] [first_ms = 0]
 *
 * Prints one line per PRN found: Doppler bin, code phase in samples (8 x byte offset + replica bit shift), peak energy and
 * its ratio to the median peak of all (PRN, Doppler) pairs. */
#include <stdio.h>
#include <stdlib.h>

#include "gpsx.h"

#define N_PRN 32
#define N_DOPP 21

static int by_value(const void *a, const void *b)
{
  const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}

int main(int argc, char **argv)
{
  if (argc < 2) {
    fprintf(stderr, "usage: %s capture.bin [n_ms] [first_ms]\n", argv[0]);
    return 2;
  }
  const int n_ms = argc > 2 ? atoi(argv[2]) : 4;
  const long first = argc > 3 ? atol(argv[3]) : 0;
  if (n_ms < 1 || n_ms > 128 || first < 0) {
    fprintf(stderr, "n_ms must be 1..128\n");
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  uint8_t *blocks = malloc((size_t)n_ms * GPSX_BYTES_PER_MS);
  if (fseek(f, first * GPSX_BYTES_PER_MS, SEEK_SET) != 0 ||
      fread(blocks, GPSX_BYTES_PER_MS, (size_t)n_ms, f) != (size_t)n_ms) {
    fprintf(stderr, "%s: fewer than %d blocks from block %ld on\n", argv[1], n_ms, first);
    return 2;
  }
  fclose(f);

  gpsx_ctx *gx = NULL;
  int rc = gpsx_create(&gx, 0, NULL);
  if (rc != GPSX_OK) {   /* no GPU: there is no CPU path to fall back to */
    fprintf(stderr, "gpsx_create: %s\n", gpsx_strerror(rc));
    return 1;
  }
  uint8_t prns[N_PRN];
  for (int i = 0; i < N_PRN; i++)
    prns[i] = (uint8_t)(i + 1);
  gpsx_acq_grid_t g = {0};
  g.n_search = 1;
  g.n_ms = n_ms;
  g.search_stride_blocks = n_ms;
  g.n_prn = N_PRN;
  g.prns = prns;
  g.dopp_min_hz = -5000;
  g.dopp_step_hz = 500;
  g.n_dopp = N_DOPP;
  g.phase_mode = GPSX_PHASES_FINE;
  g.win_start = 0;
  g.win_stop = GPSX_PHASES_BYTE;
  gpsx_peak_t *peaks = malloc(gpsx_acq_peaks_count(&g) * sizeof *peaks);
  int64_t *keys = malloc(gpsx_acq_keys_count(&g) * sizeof *keys);
  rc = gpsx_acq_grid(gx, &g, blocks, n_ms, peaks, keys);
  if (rc != GPSX_OK) {
    fprintf(stderr, "gpsx_acq_grid: %s (%s)\n", gpsx_strerror(rc), gpsx_last_error(gx));
    return 1;
  }

  /* noise floor: the median of the per-(PRN, Doppler) peak energies */
  uint32_t sorted[N_PRN * N_DOPP];
  for (int i = 0; i < N_PRN * N_DOPP; i++)
    sorted[i] = gpsx_key_energy(keys[i]);
  qsort(sorted, N_PRN * N_DOPP, sizeof sorted[0], by_value);
  const double floor_e = sorted[N_PRN * N_DOPP / 2] > 0 ? (double)sorted[N_PRN * N_DOPP / 2] : 1.0;
  printf("hypotheses=%ld noise_floor=%.0f\n", (long)N_PRN * N_DOPP * GPSX_PHASES_FINE * n_ms, floor_e);
  for (int p = 0; p < N_PRN; p++) {
    int best = 0;
    for (int d = 1; d < N_DOPP; d++)
      if (keys[p * N_DOPP + d] > keys[p * N_DOPP + best])
        best = d;
    const int64_t k = keys[p * N_DOPP + best];
    const double ratio = gpsx_key_energy(k) / floor_e;
    if (ratio >= 1.6)
      printf("PRN=%d doppler_hz=%d code_phase_samples=%u energy=%u ratio=%.2f\n", prns[p], g.dopp_min_hz + best * g.dopp_step_hz,
             gpsx_key_fine_phase(k), gpsx_key_energy(k), ratio);
  }
  free(keys);
  free(peaks);
  free(blocks);
  gpsx_destroy(gx);
  return 0;
}
