#define FVSRN_CD 1
#define FVSRN_PART 5
#include "kernels_inst.inc"
