/* oracle/hif_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see hif_oracle.h).
 * Instantiates the restatement for double (orc_d_*) and double _Complex (orc_z_*). */
#include "hif_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define T double
#define FN(x) orc_d_##x
#define IS_CPLX 0
#define CONJ(x) (x)
#define ABS(x) fabs(x)
#include "hif_oracle_impl.inc"
#undef T
#undef FN
#undef IS_CPLX
#undef CONJ
#undef ABS
#undef ORC_DUP

#define T double _Complex
#define FN(x) orc_z_##x
#define IS_CPLX 1
#define CONJ(x) conj(x)
#define ABS(x) cabs(x)
#include "hif_oracle_impl.inc"
