// ./Tahoe <model> <data> -- the reference's command line (main.cu:7-96) on top of libtahoe_amd.so.
// The reference's analytic cost model (main.cu:22-80) is replaced by the library's own strategy selector
// (TAHOE_STRATEGY_AUTO); like the reference, the program reports whether the selector agreed with the
// fastest strategy measured by SetUp, and the speedup over the baseline.
#include <iostream>

#include "BaseTahoeTest.h"

int main(int argc, char *argv[])
{
    if (argc != 3) {
        printf("Please use proper inputs: ./Tahoe [Model_Path] [Data_Path];");
        return 2;
    }
    printf("Model: %s , Data: %s\n", argv[1], argv[2]);
    BaseTahoeTest *pTest = new BaseTahoeTest(argv[1], argv[2]);
    float speedup = 0.0f;
    const int best_by_run = pTest->SetUp(speedup);
    // SetUp numbers strategies 1..5 = DIRECT, ROWTILE, TILEBLOCK, TILERING, QRING, which are the library's 1..5
    std::cout << "Performance model choose #" << pTest->auto_strategy << " strategy." << std::endl;
    if (pTest->auto_strategy == best_by_run)
        std::cout << "Performance model predicts correctly" << std::endl;
    else
        std::cout << "Performance model predicts incorrectly" << std::endl;
    std::cout << "Tahoe brings " << speedup << "x speedup." << std::endl;
    // beyond the reference's protocol: TAHOE_RESULT_JSON=<file> (machine-readable summary of this run),
    // TAHOE_LEAF_DUMP=<file> (uint32 leaf index of every (row, tree), reference heap numbering)
    if (const char *path = getenv("TAHOE_RESULT_JSON"))
        if (!pTest->write_result_json(path, best_by_run, speedup)) fprintf(stderr, "cannot write %s\n", path);
    if (const char *path = getenv("TAHOE_LEAF_DUMP"))
        if (!pTest->dump_leaf_indices(path)) fprintf(stderr, "cannot write %s: %s\n", path, tahoe_last_error());
    pTest->Free();
    delete pTest;
    return 0;
}
