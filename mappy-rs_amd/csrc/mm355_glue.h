#pragma once
#include "mm355_host.h"
