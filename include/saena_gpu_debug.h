/*
 * saena_gpu_debug.h -- test, rehearsal and bench scaffolding exported by libsaena_amd.so.
 *
 * NOT part of the drop-in boundary (that is include/saena_gpu.h): nothing a Saena maintainer binds
 * lives here.  These entry points exist so that the multi-rank code of the library can be validated
 * on ONE GPU (RCCL refuses several ranks on one device) and so that bench.py can keep a measured line
 * when an optional later leg dies.  tests/, bench.py and the perf scripts are the only callers.
 */
#ifndef SAENA_GPU_DEBUG_H
#define SAENA_GPU_DEBUG_H

#include "saena_gpu.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Halo path on a single GPU: sgpu_debug_pack runs the pack kernel (saena_matrix_matvec.cpp:25-26) and
 * downloads the send buffer; sgpu_debug_inject_halo uploads the receive buffer and makes the following
 * applies use the remote part without any exchange.  Tests route the buffers between operators on the host. */
int sgpu_debug_pack(sgpu_op *op, const value_t *v, value_t *send_host);
int sgpu_debug_inject_halo(sgpu_op *op, const value_t *recv_host);
/* An operator with remote entries applied in a context that has neither a communicator nor a host transport nor
 * an injected halo is an error (SGPU_ERR_STATE): the result would silently miss the remote part.  allow != 0
 * lifts that for this operator: only its local part is applied (plan/launch tests whose numbers mean nothing). */
int sgpu_debug_allow_local_only(sgpu_op *op, int allow);
/* diagnostic: time of the x[col] gather alone over the local part, mode 0 = production lane mapping (4 consecutive
 * nnz per lane), 1 = 64 consecutive nnz per gather instruction */
int sgpu_debug_gather_probe(sgpu_op *op, int mode, const value_t *x, int reps, float *ms);

/* bench.py (configs[4], "stresses load-balance of wavefront CSR"): the spread of work over the row blocks of the tile kernels' plan
 * (big = 0: 16 KiB tiles of <= 2048 products / 256 rows, 1: 32 KiB tiles).  out[0..7] = blocks, fewest / most entries in a block,
 * entries in all, fewest / most rows in a block, rows longer than the tile (a block of their own: the long-row path), longest row. */
int sgpu_debug_block_plan(const sgpu_op *op, int big, long *out);

/* bench.py: the streaming ceiling of a byte mix on this device -- a kernel that reads `read_bytes` with wave-coalesced 16-byte
 * loads and writes `write_bytes` with 8-byte stores and does nothing else (no gathers, no LDS), `reps` launches back to back after
 * 3 warm-up launches, plain and non-temporal loads / stores: *us = the fastest of the four forms, *mode = which (bit 0: non-temporal
 * loads, bit 1: non-temporal stores), *bytes_moved = the bytes one launch actually moves (the read stream is rounded to whole
 * 16-byte loads per written double).  A sweep over an operator that stores those bytes cannot be faster: time(ceiling) /
 * time(kernel) is `roofline.frac_of_measured`, <= 1 whether the working set is cache- or HBM-resident. */
int sgpu_debug_stream_ceiling(size_t read_bytes, size_t write_bytes, int reps, float *us, int *mode, size_t *bytes_moved);

/* Host-routed transport (validation without one GPU per rank): the context of rank `rank` of `nranks` is created
 * WITHOUT an RCCL communicator; every halo exchange is staged through host memory and handed to `exchange`, every
 * scalar reduction to `allreduce_sum`.  All library code above the transport (plans, interior/boundary kernels,
 * V-cycle, solve*, coarse levels agglomerated onto fewer ranks ...) is the multi-rank code.  exchange: send/recv are
 * packed host buffers of `elem_bytes`-sized elements, peers in ascending rank order. */
typedef int (*sgpu_host_exchange_fn)(void *user, const void *send, const int *send_rank, const int *send_count, int nsend,
                                     void *recv, const int *recv_rank, const int *recv_count, int nrecv, int elem_bytes);
typedef int (*sgpu_host_allreduce_fn)(void *user, double *v, int n);
int sgpu_debug_init_host_transport(int device_id, int rank, int nranks, sgpu_host_exchange_fn exchange,
                                   sgpu_host_allreduce_fn allreduce_sum, void *user);

/* bench.py only: from now on a fatal signal in this process (SIGSEGV/SIGBUS/SIGABRT/SIGFPE/SIGILL -- the HIP runtime
 * aborts on a GPU fault) writes `line` (may be empty) to stdout, a one-line reason to stderr, and ends the process
 * with status 128 + signal: the line measured before an optional later leg is kept, and the failure still reaches
 * the launcher as a failure.  SIGTERM is not trapped.  NULL restores the default handlers.  `line` is copied. */
int sgpu_debug_on_fatal_print(const char *line);

/* the exchange chain sgpu_init measured on the communicator (pack -> grouped send/recv with the neighbouring rank -> a kernel
 * on the received data; microseconds, the maximum over the ranks); 0 without a communicator.  The agglomeration of coarse
 * levels and the one- / two-stream thresholds of a multi-rank apply start from it. */
int sgpu_debug_chain_us(double *us);
/* number of kernel launches + graph launches + RCCL group calls the library has enqueued since sgpu_init
 * (tests: "fewer launches per V-cycle"); counts host-side enqueues, not GPU work */
int sgpu_debug_launch_count(long *launches);

/* one line describing the device of the context (name, architecture, compute units, clocks, L2, memory), for bench
 * lines and logs: the same kernel ran 1 055-1 213 us on different boxes of one pool.  `buf` receives at most len-1
 * characters and a terminator. */
int sgpu_debug_device_info(char *buf, int len);

#ifdef __cplusplus
}
#endif
#endif /* SAENA_GPU_DEBUG_H */
