#define FVSRN_CD 6
#define FVSRN_PART 1
#include "kernels_inst.inc"
