#define FVSRN_CD 6
#define FVSRN_PART 6
#include "kernels_inst.inc"
