/* cocons_hip_diag.h -- diagnostic entry points of libcocons_hip.so.
 *
 * NOT part of the drop-in boundary (include/cocons_hip.h): nothing here replaces a reference interface and no
 * caller of the package needs it: the pointwise evaluation of the device Matern routine that the parity tests pin
 * against mpmath, the run-time schedule switches, the per-task trace and the counter replay of the persistent launch.
 * (The bare-instruction probes of the fp64 pipes live in a library of their own since round 5: cocons_hip_probes.h,
 * libcocons_hip_probes.so -- the product library contains none of their kernels.)  Same conventions as the product
 * header: extern "C", host pointers, int status, no HIP call at load time.
 */
#ifndef COCONS_HIP_DIAG_H
#define COCONS_HIP_DIAG_H

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: the device routine that replaces boost::math::cyl_bessel_k + tgamma + pow at
 * src/cocons_full.cpp:293-297 (:301-305 for u >= 706), evaluated pointwise:
 * out[i] = 2^(1-nu_i)/Gamma(nu_i) * u_i^nu_i * K_nu_i(u_i).  Pinned against the mpmath grid in
 * tests/golden/besselk_grid.json (host arrays in, host array out).                    */
int cocons_debug_matern(int n, const double *nu, const double *u, double *out);

/* Schedule switches of the factorisation, settable at run time (the library reads the COCONS_* environment variables
 * of the same meaning once per process; DESIGN.md section 6 lists them): "engine", "engine_block0", "engine_pair", "panel_fused", "panel_follow",
 * "panel_diag", "panel_split", "potrf_follow", "upd_dynamic", "upd_waves", "w8_max_tiles", "dag", "dag_min_tiles", "dag_split",
 * "dag_lead" / "dag_lead2" / "dag_lead3", "dag_xcc_quota", "dag_xcd", "dag_order", "dag_bw", "dag_bh", "dag_trace" (and, for the tests, "gate_sabotage", "host_delay_us",
 * "host_delay_tile", "engine_in_wait_ms", "dag_xcd_min_quota").
 * For timing variants in alternation inside one process (tools/ab_modes.py); results do not depend on them beyond the
 * rounding of a different summation order.                                                                        */
int cocons_debug_tune(const char *name, int value);

/* Per-task time stamps of the last dependency-driven factorisation (cocons_debug_tune("dag", 1) and ("dag_trace", 1)) of a
 * fit handle: steps_out = nsteps x 16 ints (base, near, tpos, nT, H, W, tj0, k0, K, nstrip, two, need, nd_next, split, p2, p3), stamps_out =
 * ntasks x 4 ticks of the 100 MHz clock (drawn, inputs complete, product done, stored).  Returns ntasks; null outputs: sizes only. */
struct cocons_fit;
long long cocons_debug_dag_trace(struct cocons_fit *fit, int *nsteps_out, int *steps_out, unsigned long long *stamps_out,
                                 unsigned long long *engine_out);   /* engine_out (may be null): 8 stamps per pair of tiles, room for 8 (nt + 2) */


/* the first `count` task words of the last dependency-driven factorisation of the handle (host array out): [0] the one task counter,
 * [8..14] the record of a wait that ran out, [16..23] workgroups that took part per XCD, [32..39] the XCDs' own task counters */
int cocons_debug_dag_words(struct cocons_fit *fit, int count, unsigned *out);

/* host time spent ENQUEUEING evaluations on this handle and its batch slots: out2[0] = mean microseconds per evaluation, out2[1] = evaluations */
int cocons_debug_host_enqueue(struct cocons_fit *fit, double *out2);

/* The covariance assembly of an evaluation alone, `reps` times back to back: ms_out[0] = mean milliseconds per assembly
 * (tools/diag/overlap_probe.py: one handle assembling while another thread's handle factorises). */
int cocons_debug_assembly_loop(struct cocons_fit *fit, const double *theta, int reps, double *ms_out);

/* The persistent launch of the dependency-driven schedule replayed ALONE (counter passes: rocprofv3 --pmc serialises kernels,
 * and the real launch waits for the diagonal-block engine on another stream): what the engine would publish is prepared from
 * a plain-schedule factorisation of the same matrix, all hand-off words are raised, and dag_kernel runs the same task list --
 * same products, same C traffic -- between two HIP events, `reps` times.  out[0] = mean duration in ms, out[1] = the update
 * flops of the launch, out[2] = max |panel formed - plain factor| relative to max |factor| (the check), out[3] = tasks,
 * out[4] = steps.  tools/dag_replay.py.  */
int cocons_debug_dag_replay(struct cocons_fit *fit, const double *theta, const double *mean, int reps, double *out5);

#ifdef __cplusplus
}
#endif
#endif /* COCONS_HIP_DIAG_H */
