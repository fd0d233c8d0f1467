#include "mm355_glue.h"
