#define FVSRN_CD 4
#define FVSRN_PART 4
#include "kernels_inst.inc"
