#define FVSRN_CD 4
#define FVSRN_PART 0
#include "kernels_inst.inc"
