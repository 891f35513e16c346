// stand-in for <GL/gl.h>: nothing of it is used on the paths oracle/ref_glue_system.cpp runs (test infrastructure only)
#pragma once
#define GL_BGR 0x80E0
#define GL_UNSIGNED_BYTE 0x1401
#define GL_RGB 0x1907
#define GL_LUMINANCE 0x1909
