#define FVSRN_CD 2
#define FVSRN_PART 2
#include "kernels_inst.inc"
