#define FVSRN_CD 7
#define FVSRN_PART 6
#include "kernels_inst.inc"
