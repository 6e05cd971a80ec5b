/* Minimal C client of include/unikmer_hip.h (plain C99, no HIP headers needed):
 * count -k K -K -s on two sequences, then union / inter of the two sorted sets.
 *   gcc -std=c99 -Iinclude examples/count_union.c -Lunikmer_amd -lunikmer_hip -Wl,-rpath,$PWD/unikmer_amd -o count_union
 * This is what a cgo shim does: host buffers in, host buffers out. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "unikmer_hip.h"

static void die(const char *what) {
    fprintf(stderr, "%s: %s\n", what, ukm_last_error());
    exit(1);
}

/* distinct canonical k-mers of one sequence, sorted (count.go:314-436 + -s) */
static uint64_t count_sorted(ukm_ctx *c, const char *seq, int k, uint64_t *out, uint64_t cap) {
    uint64_t off[2] = {0, (uint64_t)strlen(seq)};
    uint64_t n = 0, nu = 0;
    uint64_t *codes = (uint64_t *)malloc(sizeof(uint64_t) * (off[1] + 1));
    if (ukm_encode_kmers(c, (const uint8_t *)seq, off, 1, k, 1, 0, codes, off[1], &n) != UKM_OK) die("ukm_encode_kmers");
    if (ukm_sort_u64(c, codes, n, 2 * k) != UKM_OK) die("ukm_sort_u64");
    if (ukm_unique(c, codes, NULL, n, UKM_UNIQUE, out, NULL, cap, &nu) != UKM_OK) die("ukm_unique");
    free(codes);
    return nu;
}

int main(void) {
    const char *s1 = "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGCTAGCTAGGATCCGATCGTTAGC";
    const char *s2 = "TTGCAAGGCTTAACCGGTTACGTTTTTTTTGATCGGCTAGCTAGGATCC";
    const int k = 11;
    int ndev = 0;
    ukm_ctx *c = NULL;
    uint64_t a[64], b[64], u[128], i2[64];
    uint64_t na, nb, nu = 0, ni = 0;

    if (ukm_device_count(&ndev) != UKM_OK || ndev == 0) { fprintf(stderr, "no HIP device: %s\n", ukm_last_error()); return 2; }
    if (ukm_ctx_create(0, &c) != UKM_OK) die("ukm_ctx_create");
    na = count_sorted(c, s1, k, a, 64);
    nb = count_sorted(c, s2, k, b, 64);
    if (ukm_setop2(c, UKM_OP_UNION, a, NULL, na, b, NULL, nb, 0, u, NULL, 128, &nu) != UKM_OK) die("union");
    if (ukm_setop2(c, UKM_OP_INTER, a, NULL, na, b, NULL, nb, 0, i2, NULL, 64, &ni) != UKM_OK) die("inter");
    printf("k=%d |A|=%llu |B|=%llu |A u B|=%llu |A n B|=%llu\n", k, (unsigned long long)na, (unsigned long long)nb,
           (unsigned long long)nu, (unsigned long long)ni);
    ukm_ctx_destroy(c);
    return (nu + ni == na + nb) ? 0 : 1;
}
