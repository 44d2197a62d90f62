#include "tonal_common.h"
