/* Minimal C client of include/unikmer_hip.h (plain C99, no HIP headers needed):
 * count -k K -K -s on two sequences, then union / inter of the two sorted sets.
 *   gcc -std=c99 -Iinclude examples/count_union.c -Lunikmer_amd -lunikmer_hip -Wl,-rpath,$PWD/unikmer_amd -o count_union
 * DEVICE-RESIDENT, as INTEGRATION.md's primary path: each sequence is uploaded once (page-locked staging +
 * the asynchronous transfer stream), the k-mer sets are built and stay in HBM (ukm_dev_alloc), union / inter
 * chain on them there, and only the two result sizes (and a few codes) come back. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "unikmer_hip.h"

static void die(const char *what) {
    fprintf(stderr, "%s: %s\n", what, ukm_last_error());
    exit(1);
}
#define CK(call, what) do { if ((call) != UKM_OK) die(what); } while (0)

typedef struct {
    uint64_t *codes; /* device pointer */
    uint64_t n;
} dev_set;

/* distinct canonical k-mers of one sequence, sorted (count.go:314-436 + -s); everything after the upload
 * happens on the device */
static dev_set count_sorted(ukm_ctx *c, const char *seq, int k) {
    const uint64_t len = (uint64_t)strlen(seq);
    uint64_t off[2] = {0, len};
    uint64_t n = 0;
    void *h_seq = NULL, *d_seq = NULL, *d_off = NULL, *d_codes = NULL, *d_set = NULL;
    dev_set s;
    CK(ukm_host_alloc(c, len, &h_seq), "ukm_host_alloc");
    memcpy(h_seq, seq, len);
    CK(ukm_dev_alloc(c, len, &d_seq), "ukm_dev_alloc");
    CK(ukm_dev_alloc(c, sizeof off, &d_off), "ukm_dev_alloc");
    CK(ukm_dev_alloc(c, 8 * len, &d_codes), "ukm_dev_alloc");
    CK(ukm_dev_alloc(c, 8 * len, &d_set), "ukm_dev_alloc");
    CK(ukm_copy_async(c, d_seq, h_seq, len), "ukm_copy_async"); /* returns at once */
    CK(ukm_copy(c, d_off, off, sizeof off), "ukm_copy");
    CK(ukm_copy_fence(c), "ukm_copy_fence");                    /* kernels below start after the upload */
    CK(ukm_encode_kmers(c, (const uint8_t *)d_seq, (const uint64_t *)d_off, 1, k, 1, 0, (uint64_t *)d_codes, len, &n),
       "ukm_encode_kmers");
    CK(ukm_sort_u64(c, (uint64_t *)d_codes, n, 2 * k), "ukm_sort_u64");
    CK(ukm_unique(c, (const uint64_t *)d_codes, NULL, n, UKM_UNIQUE, (uint64_t *)d_set, NULL, len, &s.n), "ukm_unique");
    s.codes = (uint64_t *)d_set;
    CK(ukm_copy_sync(c), "ukm_copy_sync");
    ukm_host_free(c, h_seq);
    ukm_dev_free(c, d_seq);
    ukm_dev_free(c, d_off);
    ukm_dev_free(c, d_codes);
    return s;
}

int main(void) {
    const char *s1 = "ACGTTGCAAGGCTTAACCGGTTACGATCGATCGGCTAGCTAGGATCCGATCGTTAGC";
    const char *s2 = "TTGCAAGGCTTAACCGGTTACGTTTTTTTTGATCGGCTAGCTAGGATCC";
    const int k = 11;
    int ndev = 0;
    ukm_ctx *c = NULL;
    dev_set a, b;
    void *d_u = NULL, *d_i = NULL;
    uint64_t nu = 0, ni = 0, first[2] = {0, 0};

    if (ukm_device_count(&ndev) != UKM_OK || ndev == 0) { fprintf(stderr, "no HIP device: %s\n", ukm_last_error()); return 2; }
    CK(ukm_ctx_create(0, &c), "ukm_ctx_create");
    a = count_sorted(c, s1, k);
    b = count_sorted(c, s2, k);
    CK(ukm_dev_alloc(c, 8 * (a.n + b.n), &d_u), "ukm_dev_alloc");
    CK(ukm_dev_alloc(c, 8 * a.n, &d_i), "ukm_dev_alloc");
    /* device pointers in, device pointers out: nothing but the counts crosses PCIe */
    CK(ukm_setop2(c, UKM_OP_UNION, a.codes, NULL, a.n, b.codes, NULL, b.n, 0, (uint64_t *)d_u, NULL, a.n + b.n, &nu), "union");
    CK(ukm_setop2(c, UKM_OP_INTER, a.codes, NULL, a.n, b.codes, NULL, b.n, 0, (uint64_t *)d_i, NULL, a.n, &ni), "inter");
    if (nu >= 2) CK(ukm_copy(c, first, d_u, sizeof first), "ukm_copy");
    printf("k=%d |A|=%llu |B|=%llu |A u B|=%llu |A n B|=%llu\n", k, (unsigned long long)a.n, (unsigned long long)b.n,
           (unsigned long long)nu, (unsigned long long)ni);
    ukm_dev_free(c, a.codes);
    ukm_dev_free(c, b.codes);
    ukm_dev_free(c, d_u);
    ukm_dev_free(c, d_i);
    ukm_ctx_destroy(c);
    return (nu + ni == a.n + b.n && (nu < 2 || first[0] < first[1])) ? 0 : 1;
}
