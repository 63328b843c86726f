/*
 * quflow_hip.h -- C ABI of libquflow_hip.so, the MI355X (gfx950) implementation of
 * quflow's isospectral hot path:  W' = (1/hbar)[P, W],  Delta P = W  on su(N).
 *
 * The reference (klasmodin/quflow) is pure Python and has NO C/FFI boundary; its
 * plug points are Python call protocols.  Each entry point
 * below therefore names the reference *Python* interface it stands under; the
 * ctypes binding a maintainer adds on the reference side is in INTEGRATION.md and
 * the in-repo mirror of those protocols is quflow_amd/ (Python).
 *
 * Conventions
 *   - every function returns int: 0 = ok, non-zero = error; the message is
 *     available from qf_last_error() (thread-local).  Nothing throws across the ABI.
 *   - matrices are N x N, C order (row major), complex128 as interleaved
 *     (re, im) doubles -- exactly numpy's layout, so host buffers are passed as-is.
 *   - the caller owns host memory; the ctx owns device memory and a HIP stream.
 *     No host pointer is retained after a call returns.
 *   - one ctx is used by one host thread at a time; independent ctxs may run
 *     concurrently -- from separate processes (one per GPU) and from separate host
 *     threads of one process, several ctxs per GPU included (thread-local error text,
 *     per-ctx stream and control state; tests/test_hip_envelope.py proves it: two
 *     threads, N = 512 and N = 1024, bit-identical to the sequential runs).
 *   - N: 2 <= N <= 8192 (qf_ctx_create refuses anything else); every size class in
 *     that range is tested against the CPU oracle.
 *   - there is NO CPU fallback: every compute entry point fails with
 *     QF_ERR_NO_DEVICE when no HIP device is present.
 */
#ifndef QUFLOW_HIP_H
#define QUFLOW_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define QF_OK 0
#define QF_ERR_INVALID 1    /* bad argument */
#define QF_ERR_NO_DEVICE 2  /* no HIP device / device index out of range */
#define QF_ERR_HIP 3        /* a HIP runtime call failed */
#define QF_ERR_STATE 4      /* call sequence violated */
#define QF_ERR_CALLBACK 5   /* a host hook of qf_isomp_hooked / qf_erk_hooked returned non-zero */
#define QF_ERR_UNSUPPORTED 6 /* a combination the reference itself rejects (NotImplementedError) */
#define QF_ERR_NONFINITE 7  /* the residual a stepper's exit test looks at is inf or NaN: the reference's scipy.linalg.norm
                             * raises ValueError("array must not contain infs or NaNs") there (isospectral.py:534, check_finite).
                             * The call is closed ON THE DEVICE at that iteration (no later launch runs, nothing is written
                             * into W): the state is left as it was after the last completed step, as the reference's array is.
                             * (Residual entries above 1.3e154 count as non-finite here -- their squares overflow --, the
                             * reference's scaled abs() gives up at 1.8e308.) */

#define QF_VERSION 100      /* 0.1.0, tracks quflow.__version__ (quflow/__init__.py:18) */

typedef struct qf_ctx qf_ctx;

/* ---- library ---------------------------------------------------------- */
int qf_version(void);
const char *qf_last_error(void);
/* number of visible HIP devices (0 when there is none; never an error) */
int qf_device_count(void);

/* ---- context: mirrors IsompCUDA.__init__(N, dtype) / DiagTriDiagOp.__init__ of the
 *      reference's device precedent (quflow/experimental/isospectral_cuda.py:52-80,
 *      quflow/experimental/cuda.py:240-354): all device buffers, the factor tables of
 *      the tridiagonal Laplacian and the stream are created once per (device, N). --- */
int qf_ctx_create(int N, int device, qf_ctx **out);
int qf_ctx_destroy(qf_ctx *ctx);
int qf_ctx_size(const qf_ctx *ctx);                 /* N */
int qf_sync(qf_ctx *ctx);                           /* hipStreamSynchronize on the ctx stream */

/* ---- geometry: quflow/geometry.py:7-9  hbar(N) = 2/sqrt(N^2-1) ---------------- */
double qf_hbar(int N);

/* ---- Laplacian backend module protocol (quflow/laplacian/__init__.py:1,
 *      tests/test_laplacian.py:134-152,226-252) -------------------------------- */

/* laplacian(N, bc): quflow/laplacian/cpu.py:55-95,604-625.  Writes the (N,N,2) float64
 * coefficient table to host memory (computed by a device kernel). */
int qf_laplacian_table(qf_ctx *ctx, int bc, double *lap_host);

/* solve_poisson(W): quflow/laplacian/cpu.py:681-734 -> _solve_cpu_skewh (cpu.py:281-362)
 * when skewh != 0, _solve_cpu_nonskewh (cpu.py:200-278) when skewh == 0
 * (select_skewherm, cpu.py:563-591).  Host in, host out. */
int qf_solve_poisson(qf_ctx *ctx, const void *W_host, void *P_host, int skewh);

/* laplace(P): quflow/laplacian/cpu.py:628-669 -> _dot_cpu_generic (cpu.py:98-108). */
int qf_laplace(qf_ctx *ctx, const void *P_host, void *W_host);

/* _solve_cpu(lap, W, P, ...) with a caller-supplied (N,N,2) coefficient table: the
 * solver underneath solve_heat / solve_helmholtz / solve_viscdamp (cpu.py:737-943).
 * The factorisation of `lap_host` is cached in the ctx under `table_key` (any
 * non-zero caller-chosen id; pass 0 to refactor on every call).
 * W_host == P_host == NULL: the solve is applied to the context's resident state in place,
 * W <- T^-1 W, asynchronously -- a `strang_splitting` half step (solve_viscdamp with theta = 1,
 * isospectral.py:466-467, 598-599) between qf_isomp / qf_isomp_continue calls without PCIe. */
int qf_solve_tridiagonal(qf_ctx *ctx, const double *lap_host, unsigned long long table_key,
                         const void *W_host, void *P_host, int skewh);
/* The factor cache is bounded: least-recently-used entries are recycled once
 * QUFLOW_HIP_FACTOR_CACHE_MB (default 512) of device memory is reached, and a key that returns
 * with a different table (fingerprint of 4096 sampled entries) is refactored.  Entries / bytes held: */
int qf_factor_cache_stats(qf_ctx *ctx, int *entries, unsigned long long *device_bytes);

/* ---- stepper protocol: isomp_fixedpoint (quflow/integrators/isospectral.py:338-613),
 *      called by simulation.solve (quflow/simulation.py:788) ---------------------- */
int qf_upload_W(qf_ctx *ctx, const void *W_host);     /* host -> ctx state W */
int qf_download_W(qf_ctx *ctx, void *W_host);         /* ctx state W -> host */

typedef struct qf_isomp_stats {
    long long total_iterations;   /* isospectral.py:426,478 */
    long long number_of_maxit;    /* steps that exhausted maxit, isospectral.py:427,540 */
    double tol_used;              /* tol actually applied (tol_auto when tol < 0), :440-452 */
    double last_resnorm;          /* residual of the last iteration performed, :534 */
} qf_isomp_stats;

/* Advances the ctx state W by `steps` isospectral-midpoint steps of length dt with the
 * built-in Hamiltonian P = Delta^-1 W (hamiltonian=solve_poisson, isospectral.py:341).
 *   tol   < 0  -> 'auto' (isospectral.py:440-452);  minit >= 1, maxit >= minit (:400-401)
 *   compsum    -> Kahan-compensated W update (:553-586), tolerance eps instead of sqrt(eps)
 *   reinitialize -> zero the iteration vector dW at every step (:471-472)
 * dW is zeroed at entry of every call (:430), so chunked calls behave like the reference.
 * Everything stays on the device, the data-dependent exit of isospectral.py:535 included: the last
 * workgroup of an iteration's second product (or k_norm_decide in the two-kernel protocol) takes the
 * decision, launches are tagged (step, iteration) and return at once when they are not due, and the
 * host only polls an 8-byte progress word in pinned memory while it enqueues ahead. */
int qf_isomp(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit,
             int compsum, int reinitialize, qf_isomp_stats *stats_out);
/* As qf_isomp, but the iteration vector dW (and the Kahan term) of the previous qf_isomp /
 * qf_isomp_continue call on this context carries over instead of being zeroed: one call of the
 * reference spans many steps and zeroes dW once (isospectral.py:430), so a host that must act
 * between the steps -- strang_splitting (isospectral.py:466-467, 598-599) or callback (:549-550) --
 * issues the first step with qf_isomp and every further one with qf_isomp_continue (the state may
 * be re-uploaded in between: Whalf restarts as W + dW, isospectral.py:481-482). */
int qf_isomp_continue(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit,
                      int compsum, int reinitialize, qf_isomp_stats *stats_out);

/* Ensembles on one GPU: k independent trajectories, one context each (all on the same device), advanced
 * by `steps` steps in one call.  Every context runs exactly the launches qf_isomp would issue for it --
 * own Hamiltonian, own exit decisions, own statistics (stats_out[k]), results bit-identical to k separate
 * qf_isomp calls -- but their streams are fed by one host loop, so the GPU overlaps the replicas (dependent-
 * launch gaps of one are filled by another; for N < 768 two replicas' workgroups share the CUs).
 * Default stepper options only (no compsum / reinitialize); dW restarts from zero as in qf_isomp.
 * How many at once: FOUR, created back to back.  The runtime gives every stream the next hardware queue when it is first
 * used and a queue's number mod 4 is the pipe that dispatches it; two replicas on one pipe lose 40 % of their combined rate
 * (N = 512: sum 18,100 timesteps/s for k = 4, 13,000 for k = 5, 16,100 for k = 8).  The call takes any k; the
 * Python mirror passes larger ensembles four contexts at a time. */
int qf_isomp_multi(qf_ctx **ctxs, int k, double dt, int steps, double tol, int minit, int maxit,
                   qf_isomp_stats *stats_out);
/* The same for contexts that hold complex64 states (qf_c64_upload_W): each runs the float32 launches qf_c64_isomp would
 * issue for it; results bit-identical to k separate qf_c64_isomp calls. */
int qf_c64_isomp_multi(qf_ctx **ctxs, int k, double dt, int steps, double tol, int minit, int maxit, qf_isomp_stats *stats_out);

/* ---- explicit (non-isospectral) steppers on the ctx state W with the built-in Hamiltonian:
 *      euler / heun / rk4, quflow/integrators/erk.py:19-59, 62-112, 115-160 (forcing = None);
 *      rhs = bracket(P, W) = (P@W - W@P)/hbar, quflow/geometry.py:41-49.
 *      skewh: the Laplacian backend's select_skewherm flag (cpu.py:563-591) for P = Delta^-1 W;
 *      with skewh != 0 and an exactly skew-Hermitian state W@P = (P@W)^H is used. ------- */
#define QF_ERK_EULER 0
#define QF_ERK_HEUN 1
#define QF_ERK_RK4 2
int qf_erk(qf_ctx *ctx, int method, double dt, int steps, int skewh);
/* the same on a stack of k states (erk.py with W.shape = (k,N,N)): P from state 0 (cpu.py:696-697), bracket(P, W)
 * broadcast over the stack (geometry.py:41-49).  states_host: (k,N,N) complex128, overwritten with the result. */
int qf_erk_states(qf_ctx *ctx, void *states_host, int k, int method, double dt, int steps, int skewh);

/* ---- isomp_simple (isospectral.py:254-335) and isomp_quasinewton (isospectral.py:155-251) on the
 *      ctx state W, hamiltonian = solve_poisson, skew-Hermitian case.  The reference's two
 *      lu_solve calls per pass (A = I - (stepsize/2) Ptilde) are replaced by a Newton-Schulz
 *      inverse on the matrix cores (A is always well conditioned for skew-Hermitian Ptilde).
 *      qf_isomp_quasinewton: tol < 0 -> 'auto' (eps*stepsize*|W|_inf, :190-191);
 *      stats: total_iterations, number_of_maxit (steps whose loop ran out), tol_used. ------ */
int qf_isomp_simple(qf_ctx *ctx, double dt, int steps);
int qf_isomp_quasinewton(qf_ctx *ctx, double dt, int steps, double tol, int maxit, qf_isomp_stats *stats_out);
/* ... with a foreign `hamiltonian(Wtilde)` (isospectral.py:207, 286): the `hamiltonian` / `user` members of a
 * qf_isomp_hooks table (declared below).  Wtilde goes down and Ptilde comes up once per pass; the inverse, the
 * two solves and the update stay on the device.  (`forcing` is accepted and ignored by the reference's
 * isomp_simple / isomp_quasinewton -- an `assert` on an exception object, :185-186, 283-284 -- so there is no
 * forcing member to honour here.) */
struct qf_isomp_hooks;
int qf_isomp_simple_hooked(qf_ctx *ctx, double dt, int steps, const struct qf_isomp_hooks *hooks);
int qf_isomp_quasinewton_hooked(qf_ctx *ctx, double dt, int steps, double tol, int maxit, qf_isomp_stats *stats_out,
                                const struct qf_isomp_hooks *hooks);

/* ---- isomp on a stack of k states (isospectral.py:463-611 with W.shape = (k,N,N): P from state 0,
 *      the same products for every state, exit test on state 0) and, with magnetic != 0 and
 *      k == 2, magmp (quflow/integrators/mhd.py:235-456, hamiltonian = solve_mhd: P = Delta^-1 W,
 *      B = Delta Theta).  states_host: (k,N,N) complex128, overwritten with the result.
 *      tol < 0 -> 'auto' (sqrt(eps)*dt/hbar*|state 0|_inf). ------------------------------- */
int qf_isomp_states(qf_ctx *ctx, void *states_host, int k, double dt, int steps, double tol, int minit, int maxit,
                    int reinitialize, int magnetic, qf_isomp_stats *stats_out);

/* ---- the stepper with HOST HOOKS: isomp_fixedpoint's `forcing`, foreign `hamiltonian`, `strang_splitting`,
 *      `callback` (isospectral.py:338-353, 403-423, 466-467, 488-492, 512-520, 547-551, 598-603), the
 *      general branch of select_skewherm(False) (:504-505), and all of them -- with compsum -- on
 *      (k,N,N) stacks (P from the Hamiltonian, exit test on state 0, :527-532).  The state, dW, Whalf,
 *      the products and the Kahan term stay on the device; a hook receives / fills pinned host matrices
 *      owned by the library (valid during the call only), (k,N,N) or (N,N) complex128 as noted, and
 *      returns 0 (anything else aborts the call with QF_ERR_CALLBACK).  NULL pointers = hook absent;
 *      hamiltonian == NULL: the built-in P = Delta^-1 W[0] on the device. ------------------------------ */
typedef struct qf_isomp_hooks {
    void *user;
    /* P (N,N) = hamiltonian(Whalf (k,N,N)[, time = t + dt/2 when hamiltonian_takes_time]) */
    int (*hamiltonian)(void *user, const void *Whalf, void *P, double time);
    /* F (k,N,N) = forcing(P (N,N), Whalf (k,N,N)[, time = t + dt/2 when forcing_takes_time]) */
    int (*forcing)(void *user, const void *P, const void *Whalf, void *F, double time);
    /* W (k,N,N) <- strang_splitting(h, W), in place, h = dt/2, before and after every step */
    int (*strang)(void *user, double h, void *W);
    /* callback(W, dW2): the state before the step's update and 2 (PW - PW^H) (k,N,N each) */
    int (*callback)(void *user, const void *W, const void *dW2);
    int hamiltonian_takes_time, forcing_takes_time;   /* the autonomy probing (:403-423) is the caller's */
    int has_time;                                     /* `time` is not None: advanced by dt per step */
    double time;
    int skewh;                                        /* integrators' select_skewherm flag (isospectral.py:96-118):
                                                         commutator as PW - PW^H or as PW - Whalf@Phalf */
    int solve_skewh;                                  /* the Laplacian backend's flag (cpu.py:563-591) for the built-in P */
    /* instead of `strang`: W <- T^-1 W with this (N,N,2) table, on the device (solve_viscdamp half step) */
    const double *strang_table;
    unsigned long long strang_key;
    /* magmp_fixedpoint (quflow/integrators/mhd.py:235-456) on a (2,N,N) state (W, Theta), k == 2: the vorticity
     * state gets the magnetic terms [B, Theta] on top.  `hamiltonian` then fills P with TWO (N,N) matrices,
     * (P, B) -- the pair solve_mhd returns (mhd.py:10-18); NULL: the built-in P = Delta^-1 W, B = Delta Theta.
     * `forcing` receives P (the first of the two) and the (2,N,N) state; `callback` as above. */
    int magnetic;
    /* stacks (qf_isomp_hooked with k > 1, qf_erk_states_hooked): 1 = `hamiltonian` fills ONE stream matrix per state of
     * the stack (k matrices: a Hamiltonian that returns a (k,N,N) array, bracket(P, W) is then a batched product), 0 = one
     * for all states; -1 = not known yet: the hook stores 0 or 1 here during its first call and the library reads the
     * field back behind that call (the reference never asks in advance, isospectral.py:488-491). */
    int states_p;
} qf_isomp_hooks;
/* states_host: (k,N,N) complex128, overwritten with the result.  compsum with forcing: QF_ERR_UNSUPPORTED (:588-589) */
int qf_isomp_hooked(qf_ctx *ctx, void *states_host, int k, double dt, int steps, double tol, int minit, int maxit,
                    int compsum, int reinitialize, const qf_isomp_hooks *hooks, qf_isomp_stats *stats_out);
/* euler / heun / rk4 (erk.py:19-160) with `forcing(P, W)` and / or a foreign `hamiltonian(W)` (the `hamiltonian`,
 * `forcing`, `user` and `skewh` members of the hook table; k = 1).  W_host: (N,N), overwritten. */
int qf_erk_hooked(qf_ctx *ctx, void *W_host, int method, double dt, int steps, const qf_isomp_hooks *hooks);
/* The same on a (k,N,N) stack (quflow/integrators/erk.py with batched input): `hamiltonian(stack)` fills ONE (N,N) stream
 * matrix, `forcing(P, stack)` a stack; the built-in Hamiltonian solves for state 0 (cpu.py:696-697). */
int qf_erk_states_hooked(qf_ctx *ctx, void *states_host, int k, int method, double dt, int steps, const qf_isomp_hooks *hooks);

/* ---- spherical-harmonics <-> matrix transforms (quflow/quantization.py).  The quantization
 *      basis (compute_basis, quantization.py:68-113: N(N+1)(2N+1)/6 doubles, block m row-major at
 *      basis_break_index(m, N)) is uploaded once and stays resident in HBM; a transform is one
 *      HBM-bound sweep over it.  A NULL matrix pointer means "the ctx state W" (initial data /
 *      'shr' output of a resident trajectory without moving W over PCIe). ------------------ */
int qf_basis_upload(qf_ctx *ctx, const double *basis_host, long long count);
/* compute_basis(N) (quantization.py:68-113) on the device, straight into the resident copy: the
 * eigenvectors of the tridiagonal blocks of the direct Laplacian (laplacian/direct.py:19-62) by
 * one twisted factorisation per vector at the known eigenvalues -el(el+1) (the reference calls
 * LAPACK's tridiagonal eigensolver), scaled and oriented as quantization.py:45-65,99-106. */
int qf_basis_compute(qf_ctx *ctx);
int qf_basis_download(qf_ctx *ctx, double *basis_host, long long count);
/* shr2mat_(omega, basis, W_out), quantization.py:188-245 (W_out zeroed first as in shr2mat, :474);
 * n_omega < N^2 band-limits to el < int(sqrt(n_omega)) (:204-208) */
int qf_shr2mat(qf_ctx *ctx, const double *omega_host, long long n_omega, void *W_host);
/* (omega_host == NULL in either direction: the coefficients stay on / come from the device copy
 *  the previous transform left, so that a filter "mat2shr -> scale -> shr2mat" or a timing loop
 *  never crosses PCIe) */
/* mat2shr_(W, basis, omega_out), quantization.py:286-327, omega_out zeroed first (:516) */
int qf_mat2shr(qf_ctx *ctx, const void *W_host, double *omega_host, long long n_omega);
/* shc2mat_ / mat2shc_, quantization.py:331-396: omega is N^2 complex128 */
int qf_shc2mat(qf_ctx *ctx, const void *omega_host, void *W_host);
int qf_mat2shc(qf_ctx *ctx, const void *W_host, void *omega_host);

/* ---- diagnostics on the ctx state W: quflow/physics.py:26-38 with
 *      inner_L2 (quflow/geometry.py:72-76) -------------------------------------- */
int qf_diagnostics(qf_ctx *ctx, double *energy_euler, double *enstrophy);
/* qf_isomp followed by qf_diagnostics with ONE synchronisation: the diagnostics' launches are queued behind
 * the last step (an output chunk of simulation.solve / of an ensemble rank: advance, then log energy and
 * enstrophy, quflow/simulation.py:788-803).  Same results as the two calls. */
int qf_isomp_diag(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum,
                  int reinitialize, qf_isomp_stats *stats, double *energy_euler, double *enstrophy);
/* matrix infinity norm of the state, np.linalg.norm(W, inf) (isospectral.py:448) */
int qf_norm_inf_W(qf_ctx *ctx, double *out);

/* ---- measurement support (bench.py): HIP-event timing on the ctx stream -------- */
#define QF_KERNEL_POISSON 0
#define QF_KERNEL_GEMM1 1
#define QF_KERNEL_GEMM2 2
#define QF_KERNEL_NORM 3
#define QF_KERNEL_UPDATE 4
#define QF_KERNEL_SLICE 5      /* digit slicing of the int8 products' operands (QUFLOW_HIP_GEMM=i8) */
#define QF_KERNEL_COUNT 6

/* `mask` selects kernels (bit QF_KERNEL_x); every launch of a selected hot-path kernel
 * inside qf_isomp is bracketed by a hipEvent pair on the ctx stream and qf_profile_read
 * sums the elapsed times.  mask = 0 switches the instrumentation off. */
int qf_profile_enable(qf_ctx *ctx, int mask);
int qf_profile_reset(qf_ctx *ctx);
int qf_profile_read(qf_ctx *ctx, int kernel_id, long long *launches, double *total_ms);
/* Sampling: bracket only every `stride`-th launch of an enabled kernel id (default 1 = every launch);
 * qf_profile_read then returns the measured launches and their time, qf_profile_seen all launches
 * of that id since the last reset (bench.py scales the measured mean to them). */
int qf_profile_stride(qf_ctx *ctx, int stride);
int qf_profile_seen(qf_ctx *ctx, int kernel_id, long long *seen);
/* The HIP device with ordinal `device` as this process sees it: a JSON object {"ordinal":.., "pci_bus_id":"0000:05:00.0",
 * "name":"..", "gcn_arch":"gfx950..", "compute_units":.., "memory_bytes":..} -- what a rank of a multi-GPU launch prints
 * about the device it bound (bench.py).  Returns the length of the text (as snprintf), a negative status on error. */
int qf_device_info(int device, char *buf, int n);
/* What this context LAUNCHED for each role of the hot path -- recorded by the launchers themselves at the moment of the
 * launch (not re-derived from the selection rules): a JSON object
 *   {"N":.., "laplacian_inverse":{..}, "first_product":{..}, "second_product":{..}, "slicing":{..}}
 * with, per role that has run since the context was created, the kernel, its tile, workgroups and threads, and for the
 * second product the tiles it multiplies and their share of the full grid ("tile_share"); null for a role that has not
 * run.  bench.py labels its `roofline` object with this.  Returns the length of the text (as snprintf: the text is
 * truncated to n - 1 bytes when n is smaller), a negative status on error. */
int qf_plan_describe(qf_ctx *ctx, char *buf, int n);
/* stream-ordered stopwatch: start/stop record events on the ctx stream */
int qf_timer_start(qf_ctx *ctx);
int qf_timer_stop(qf_ctx *ctx, double *elapsed_ms);

/* ---- debug / parity access to device intermediates (tests only) ---------------- */
#define QF_BUF_W 0
#define QF_BUF_DW 1
#define QF_BUF_WHALF 2
#define QF_BUF_PHALF 3
#define QF_BUF_PW 4
int qf_download_buffer(qf_ctx *ctx, int which, void *host);
/* Guard zones around every device allocation of the library (QUFLOW_HIP_DEBUG_GUARD=1 in the environment when the process
 * starts; csrc/guard.hip): reads back the zones of every live allocation and reports how many allocations were made under
 * the guard, how many zones were found damaged so far (live ones now, released ones when they were released) and a
 * description of the first one (truncated to n - 1 bytes).  All zero / empty when the variable is not set.  Stores that
 * land outside an operand are otherwise swallowed by the allocator's granularity; the GPU suite and the size sweeps are
 * run once per round under the variable (tests/conftest.py fails such a session on damage). */
int qf_debug_guard_check(long long *allocations, long long *damaged, char *first, int n);
/* n pairs (er[i], ei[i]) -> out_modulus[i] = |er + i ei| as every residual row sum of the stepper forms it (the library's
 * own square-root sequence, isospectral.py:526,534), out_sqrt[i] = the compiler's sqrt of the same argument: the two
 * must agree bit for bit in the normal range -- the iteration counts rest on it.  n <= N*N of the context. */
int qf_debug_modulus(qf_ctx *ctx, int n, const double *er_host, const double *ei_host, double *out_modulus_host,
                     double *out_sqrt_host);
/* C = A @ B for host matrices through the MFMA zgemm of the stepper (parity tests of
 * the commutator pair, isospectral.py:496,499). */
int qf_zgemm(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host);
/* The commutators of quflow/integrators/isospectral.py:22-57 on host matrices, product(s) AND combination on the device
 * (round 6: one PCIe round trip, no host pass):  skewherm = 1: X - X^H with X = W @ P (commutator_skewherm, one product);
 * skewherm = 0: W @ P - P @ W (commutator_generic, two products).  The subtraction is the reference's elementwise one
 * (x - y, x - conj(y^T): exact negation, one rounding), so the result has the bits of the product followed by numpy's
 * `VF -= ...`.  Uses the context's per-iteration staging matrices (stage, Phalf, PW, Whalf), which are free between
 * stepper calls -- the resident state W and a carried increment are not touched. */
int qf_commutator(qf_ctx *ctx, const void *W_host, const void *P_host, void *C_host, int skewherm);
/* C = A @ B on the INT8 matrix cores by digit splitting (ozaki.hip; BASELINE.json config 3's
 * low-precision-MFMA commutator): A general, B SKEW-HERMITIAN (as every right operand of the
 * iteration is); truncation error 2^-35 of (row scale of A) x (column scale of B). N % 64 == 0. */
int qf_zgemm_i8(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host);
/* The two products of ONE fixed-point iteration with their fused epilogue, on host operands
 * (isospectral.py:496-509,481-482,526-534):  PW = Phalf @ Whalf;  dW_new = PW @ Phalf + (PW - PW^H);
 * Whalf_new = W + dW_new;  rowsum[i] = sum_j |dW_old[i,j] - dW_new[i,j]|  (N doubles).
 * variant (low 4 bits) 0: full second product; 1: the upper-triangle stream-K form the stepper uses
 * for skew-Hermitian W at the large sizes (N % 64 == 0 only); 2: the upper triangle of 32 x 32 tiles (smaller sizes, any N
 * >= 64).  The parity tests also choose the partition here, which the stepper takes from rules: kind 1: bits 8-15 = least
 * K-tiles per workgroup, bits 16-23 = 64 + epilogue weight (0: the rule); kind 2: bits 8-11 / 12-15 = K pieces per
 * off-diagonal / diagonal tile (1, 2, 4; 0: the rule).  Parity-test entry: the stepper itself never round-trips through
 * the host. */
int qf_fixedpoint_products(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                           const void *dW_old_host, int variant, void *dW_new_host, void *Whalf_new_host,
                           double *rowsum_host);

/* ---- complex64 data.  The reference computes complex64 input in single precision throughout: float32
 *      coefficient tables and a float32 Thomas solve (quflow/laplacian/cpu.py:725, `dtype=type(W[0,0].real)`),
 *      complex64 np.matmul for the two products (quflow/integrators/isospectral.py:496,499), complex64
 *      elementwise passes (:481-592), and the automatic tolerance from the float32 machine epsilon
 *      (np.finfo(W.dtype).eps, :440-448).  These entry points are that path on the device: matrices are
 *      N x N, C order, complex64 as interleaved (re, im) floats; the solve runs on float32 factor tables, the
 *      products on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), Kahan summation (compsum) in float32.  The
 *      float32 working set of a context is allocated on first use; its state is separate from the complex128
 *      one.  Scalars (dt, tol, statistics, diagnostics) stay double. -------------------------------------- */
/* laplacian(N, bc, dtype=float32): cpu.py:55-95.  (N,N,2) float32 to host memory. */
int qf_c64_laplacian_table(qf_ctx *ctx, int bc, float *lap_host);
/* solve_poisson(W) for complex64 W: cpu.py:681-734 with float32 tables (:725).  Host in, host out. */
int qf_c64_solve_poisson(qf_ctx *ctx, const void *W_host, void *P_host, int skewh);
/* _solve_cpu(lap, W, P, ...) with a caller-supplied (N,N,2) float32 table on complex64 data: solve_heat /
 * solve_helmholtz / solve_viscdamp / solve_globalqg on complex64 input (cpu.py:737-943 build their tables with
 * dtype=type(W[0,0].real)).  Host in, host out; the table is factorised on every call. */
int qf_c64_solve_tridiagonal(qf_ctx *ctx, const float *lap_host, const void *W_host, void *P_host, int skewh);
/* laplace(P) for complex64 P: cpu.py:628-669 with the float32 table (:664). */
int qf_c64_laplace(qf_ctx *ctx, const void *P_host, void *W_host);
int qf_c64_upload_W(qf_ctx *ctx, const void *W_host);     /* host complex64 -> the context's complex64 state */
int qf_c64_download_W(qf_ctx *ctx, void *W_host);
/* isomp_fixedpoint on the complex64 state (arguments as qf_isomp); tol < 0 -> 'auto' with the float32 machine
 * epsilon: sqrt(eps32)*dt/hbar*|W|_inf, or eps32*... with compsum (isospectral.py:440-448). */
int qf_c64_isomp(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                 qf_isomp_stats *stats_out);
int qf_c64_isomp_continue(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                          qf_isomp_stats *stats_out);
/* energy_euler / enstrophy of the complex64 state (quflow/physics.py:26-38) */
int qf_c64_diagnostics(qf_ctx *ctx, double *energy_euler, double *enstrophy);
/* C = A @ B for complex64 host matrices through the stepper's fp32-MFMA product (parity tests; isospectral.py:496,499) */
int qf_cgemm(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host);
/* the two products of one fixed-point iteration with the fused epilogue on complex64 host operands (as
 * qf_fixedpoint_products, full second product); rowsum: N doubles */
int qf_c64_fixedpoint_products(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                               const void *dW_old_host, void *dW_new_host, void *Whalf_new_host, double *rowsum_host);
/* The same through the upper-triangle second product (N % 64 == 0, skew-Hermitian operands): the kernel the complex64
 * stepper uses from N = 768 on.  dW comes back completed by its mirror image. */
int qf_c64_fixedpoint_products_tri(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                                   const void *dW_old_host, void *dW_new_host, void *Whalf_new_host, double *rowsum_host);

/* ---- ensemble diagnostics gather over RCCL, one process per GPU (the reference
 *      has no distributed code -- this row has no reference interface to cite).  Torch-free alternative to
 *      the torch.distributed route of quflow_amd/ensemble.py: rank 0 draws the id, the ranks exchange its
 *      128 bytes themselves (quflow_amd/comm.py: a TCP hand-out on MASTER_ADDR), every rank creates its
 *      communicator.  librccl.so is dlopen'ed on first use (QUFLOW_HIP_RCCL_LIB overrides the name). ---- */
#define QF_COMM_ID_BYTES 128
typedef struct qf_comm qf_comm;
int qf_comm_unique_id(void *id128);
int qf_comm_create(qf_comm **out, int device, int nranks, int rank, const void *id128);
/* recv_host holds nranks * count doubles, rank r's block at r * count; every rank passes the same count */
int qf_comm_allgather_f64(qf_comm *comm, const double *send_host, int count, double *recv_host);
int qf_comm_barrier(qf_comm *comm);
int qf_comm_destroy(qf_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* QUFLOW_HIP_H */
