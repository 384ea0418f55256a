#define FVSRN_CD 7
#define FVSRN_PART 1
#include "kernels_inst.inc"
