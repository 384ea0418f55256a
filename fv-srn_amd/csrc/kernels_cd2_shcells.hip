#define FVSRN_CD 2
#define FVSRN_PART 6
#include "kernels_inst.inc"
