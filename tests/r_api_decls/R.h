/* TEST INFRASTRUCTURE: see Rinternals.h in this directory */
#include <string.h>
