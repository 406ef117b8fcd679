// Test probe (not product): prints what qreg_plan (tahoe_amd/csrc/qring_internal.h) decides for the argument tuples on stdin,
// one "rows cus force cost3 big slice_trees" per line -> "rows3 chains".  Built by tests/test_tile_plan.py with hipcc (host side only runs).
#include <cstdio>

#include "qring_internal.h"

int main()
{
    unsigned long long rows, cost3, big;
    int cus, force, slice_trees;
    while (scanf("%llu %d %d %llu %llu %d", &rows, &cus, &force, &cost3, &big, &slice_trees) == 6) {
        size_t rows3 = 0;
        int chains = 0;
        tahoe::qreg_plan((size_t)rows, cus, force, &rows3, &chains, (size_t)cost3, (size_t)big, slice_trees);
        printf("%zu %d\n", rows3, chains);
    }
    return 0;
}
