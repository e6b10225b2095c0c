// abi.hip -- ABI version stamp of libcmda_hip.so.
#include "common.h"
extern "C" int cmda_abi_version(void) { return 8; }
