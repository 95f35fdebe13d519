// The index, graph-build and gather-SpMM kernels as ONE translation unit: riders (common.h) may carry a recorded launch of one
// of these files inside a launch of another (the next step's hop-0 row order beside the classifier's gather-SpMM), which needs
// both kernel bodies in one compilation.  Nothing else changes: every kernel is compiled exactly as in its own file.
#define GRAPES_HOP_UNITY 1
#include "index_kernels.hip"
#include "prep_kernels.hip"
#include "spmm_kernels.hip"
