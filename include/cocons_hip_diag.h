/* cocons_hip_diag.h -- diagnostic entry points of libcocons_hip.so.
 *
 * NOT part of the drop-in boundary (include/cocons_hip.h): nothing here replaces a reference interface and no
 * caller of the package needs it.  These are the probes behind the numbers in DESIGN.md / profiles/ (what the
 * fp64 matrix pipe sustains, whether matrix and vector fp64 run side by side) and the pointwise evaluation of
 * the device Matern routine that the parity tests pin against mpmath.  Same conventions as the product header:
 * extern "C", host pointers, int status, no HIP call at load time.
 */
#ifndef COCONS_HIP_DIAG_H
#define COCONS_HIP_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

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

/* Diagnostic: the device routine that replaces boost::math::cyl_bessel_k + tgamma + pow at
 * src/cocons_full.cpp:293-297 (:301-305 for u >= 706), evaluated pointwise:
 * out[i] = 2^(1-nu_i)/Gamma(nu_i) * u_i^nu_i * K_nu_i(u_i).  Pinned against the mpmath grid in
 * tests/golden/besselk_grid.json (host arrays in, host array out).                    */
int cocons_debug_matern(int n, const double *nu, const double *u, double *out);

/* both probes at once on two streams: out4[0], out4[1] = TFLOP/s of the MFMA / the FMA kernel while the other one
 * runs, out4[2], out4[3] = their durations in ms -- do the matrix and the vector fp64 pipes run concurrently? */
int cocons_corun_probe(int bpc_mfma, int bpc_vfma, int iters_mfma, int iters_vfma, double *out4);

/* Schedule switches of the factorisation, settable at run time (the library reads the COCONS_* environment variables
 * of the same meaning once per process): "engine", "upd_dynamic", "upd_waves", "w8_max_tiles" (and, for the tests,
 * "gate_sabotage").
 * For timing variants in alternation inside one process (tools/ab_modes.py); results do not depend on them beyond the
 * rounding of a different summation order.                                                                        */
int cocons_debug_tune(const char *name, int value);

/* Per-task time stamps of the last dependency-driven factorisation (cocons_debug_tune("dag", 1) and ("dag_trace", 1)) of a
 * fit handle: steps_out = nsteps x 16 ints (base, near, tpos, nT, H, W, tj0, k0, K, nstrip, two, need, nd_next, split, p2, p3), stamps_out =
 * ntasks x 4 ticks of the 100 MHz clock (drawn, inputs complete, product done, stored).  Returns ntasks; null outputs: sizes only. */
struct cocons_fit;
long long cocons_debug_dag_trace(struct cocons_fit *fit, int *nsteps_out, int *steps_out, unsigned long long *stamps_out,
                                 unsigned long long *engine_out);   /* engine_out (may be null): 8 stamps per pair of tiles, room for 8 (nt + 2) */

#ifdef __cplusplus
}
#endif
#endif /* COCONS_HIP_DIAG_H */
