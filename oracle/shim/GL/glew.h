// stand-in for <GL/glew.h>: nothing of it is used on the paths oracle/ref_glue_system.cpp runs (test infrastructure only)
#pragma once
