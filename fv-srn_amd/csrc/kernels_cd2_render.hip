#define FVSRN_CD 2
#define FVSRN_PART 1
#include "kernels_inst.inc"
