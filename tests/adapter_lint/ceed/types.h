#include "ceed.h"
