/* cocons_hip_probes.h -- bare-instruction probes of the fp64 pipes of gfx950 (libcocons_hip_probes.so, built from
 * cocons_amd/csrc/probes.hip).  Measurement tools, not product code and not part of any boundary: what the matrix pipe
 * sustains is the ceiling the trailing-update kernel is priced against (profiles/r02_mfma_f64_probe.json).
 * extern "C", int status (0 ok, < 0 error: cocons_probe_last_error()), no HIP call at load time.
 */
#ifndef COCONS_HIP_PROBES_H
#define COCONS_HIP_PROBES_H

#ifdef __cplusplus
extern "C" {
#endif

const char *cocons_probe_last_error(void);

/* fp64 MFMA issue-rate probe: back-to-back v_mfma_f64_16x16x4_f64 on every SIMD with
 * `blocks_per_cu` 256-thread workgroups per CU; returns the sustained TFLOP/s.  Evidence
 * for the roofline peak the update kernel is priced against. */
int cocons_mfma_f64_probe(int blocks_per_cu, double *tflops);
/* Extended probe: nacc (4 / 8 / 16) independent accumulators per wave, form 0 = v_mfma_f64_16x16x4_f64,
 * 1 = v_mfma_f64_4x4x4_4b_f64, 3 = the 4x4x4 form with sixteen accumulators fed from eight DISTINCT operand
 * registers (nacc ignored); `reps` bursts of `iters` loop iterations separated by idle gaps of gap_us
 * (0 = back to back).  out4[0] = TFLOP/s inside the bursts, [1] = clock the chip held inside the kernel
 * in GHz (s_memtime / s_memrealtime), [2] = shader cycles per MFMA instruction per wave, [3] = mean
 * burst duration in ms.  Tells issue rate per clock apart from the clock the chip sustains under load. */
int cocons_mfma_f64_probe_ex(int blocks_per_cu, int nacc, int form, int iters, int gap_us, int reps, double *out4);
/* companion: independent v_fma_f64 chains -- the fp64 vector rate the chip sustains */
int cocons_vfma_f64_probe(int blocks_per_cu, double *tflops);

/* both probes at once on two streams: out4[0], out4[1] = TFLOP/s of the MFMA / the FMA kernel while the other one
 * runs, out4[2], out4[3] = their durations in ms -- do the matrix and the vector fp64 pipes run concurrently? */
int cocons_corun_probe(int bpc_mfma, int bpc_vfma, int iters_mfma, int iters_vfma, double *out4);

#ifdef __cplusplus
}
#endif
#endif /* COCONS_HIP_PROBES_H */
