#define FVSRN_CD 6
#define FVSRN_PART 2
#include "kernels_inst.inc"
