/* A plain C99 host of the C ABI (include/eigenex_hip.h is C, not C++): the GPU-free entry points -- partition, shard plan,
 * collective schedule -- called the way a cgo / JNI / ctypes binding would call them.  Built with `gcc -std=c99 -pedantic`
 * and run by tests/test_cabi_and_host_logic.py on the CPU.  Prints "c host: ok" and exits 0 when every check holds. */
#include <stdio.h>
#include <stdlib.h>

#include "eigenex_hip.h"

#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      fprintf(stderr, "FAILED %s:%d %s (%s)\n", __FILE__, __LINE__, #c, eigenex_last_error()); \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main(void) {
  /* tridiagonal matrix of 10 rows on 3 shards: shard 1 owns rows [3, 6) */
  const int64_t n = 10;
  int64_t rb = 0, re = 0;
  CHECK(eigenex_version() >= 100);
  CHECK(eigenex_partition(n, 3, 1, &rb, &re) == 0 && rb == 3 && re == 6);
  int32_t rowptr[4] = {0, 3, 6, 9};
  int32_t col[9] = {2, 3, 4, 3, 4, 5, 4, 5, 6}; /* global columns of rows 3, 4, 5 */
  eigenex_plan_t plan = NULL;
  CHECK(eigenex_plan_create(n, 3, 1, rowptr, col, &plan) == 0);
  int64_t nloc = 0, npad = 0, nnz = 0, nhalo = 0, nsend_rows = 0;
  int nrecv = 0, nsend = 0;
  CHECK(eigenex_plan_sizes(plan, &nloc, &npad, &nnz, &nhalo, &nrecv, &nsend, &nsend_rows) == 0);
  CHECK(nloc == 3 && npad == 64 && nnz == 9 && nhalo == 2 && nrecv == 2 && nsend == 0);
  int32_t halo[2], lcol[9], peer[2];
  int64_t off[2], cnt[2], contig[2];
  CHECK(eigenex_plan_halo_columns(plan, halo) == 0 && halo[0] == 2 && halo[1] == 6);
  CHECK(eigenex_plan_local_columns(plan, lcol) == 0);
  CHECK(lcol[0] == 64 && lcol[1] == 0 && lcol[2] == 1 && lcol[8] == 65); /* halo slots sit behind the padded rows */
  CHECK(eigenex_plan_recv_segments(plan, peer, off, cnt) == 0);
  CHECK(peer[0] == 0 && off[0] == 0 && cnt[0] == 1 && peer[1] == 2 && off[1] == 1 && cnt[1] == 1);
  {
    /* three rows = one 256-row tile, and it reads halo columns: no interior tile, one boundary tile */
    int32_t ti[1], tb[1];
    int64_t ni = -1, nb = -1;
    CHECK(eigenex_plan_tiles(plan, NULL, &ni, NULL, &nb) == 0 && ni == 0 && nb == 1);
    CHECK(eigenex_plan_tiles(plan, ti, &ni, tb, &nb) == 0 && tb[0] == 0);
    CHECK(eigenex_plan_tiles(NULL, ti, &ni, tb, &nb) < 0);
  }
  /* the neighbours ask for row 3 (shard 0) and row 5 (shard 2) */
  {
    const int32_t want0[1] = {3}, want2[1] = {5};
    int32_t rows[2];
    CHECK(eigenex_plan_add_request(plan, 0, want0, 1) == 0);
    CHECK(eigenex_plan_add_request(plan, 2, want2, 1) == 0);
    CHECK(eigenex_plan_sizes(plan, NULL, NULL, NULL, NULL, NULL, &nsend, &nsend_rows) == 0 && nsend == 2 && nsend_rows == 2);
    CHECK(eigenex_plan_send_segments(plan, peer, off, cnt, contig) == 0);
    CHECK(peer[0] == 0 && cnt[0] == 1 && contig[0] == 0 && peer[1] == 2 && cnt[1] == 1 && contig[1] == 2);
    CHECK(eigenex_plan_send_rows(plan, rows) == 0 && rows[0] == 0 && rows[1] == 2);
    /* a row this shard does not own is refused */
    {
      const int32_t bad[1] = {7};
      CHECK(eigenex_plan_add_request(plan, 2, bad, 1) != 0);
    }
  }
  CHECK(eigenex_plan_destroy(plan) == 0);
  /* collectives of three step calls in one batch, alpha fusion on: the third (last) call closes its own alpha */
  {
    int ops[16], counts[16], nops = 0, pending = 0, call;
    int total_allreduce = 0;
    for (call = 0; call < 3; ++call) {
      int i;
      CHECK(eigenex_lanczos_collectives(call, call == 2, &pending, 1, 0, EIGENEX_ORTHO_BATCHED, 1, 0, ops, counts, 16, &nops) == 0);
      for (i = 0; i < nops; ++i) total_allreduce += ops[i] == EIGENEX_COLL_ALLREDUCE;
    }
    CHECK(pending == 0);
    CHECK(total_allreduce == 1 /* start norm */ + 2 + 2 /* fused dots, norm per step */ + 1 /* closing alpha */);
  }
  /* malformed input comes back as an error code with a message, not as a crash */
  {
    int32_t bad_rowptr[4] = {0, 5, 2, 9};
    eigenex_plan_t p2 = NULL;
    CHECK(eigenex_plan_create(n, 3, 1, bad_rowptr, col, &p2) == EIGENEX_ERR_ARG && p2 == NULL);
  }
  printf("c host: ok\n");
  return 0;
}
