// stand-in for SiftGPU.h (the feature extractor / matcher the reference links; not present here): every call throws.  The
// front end (extractAndMatchFeaturesGPU) is outside the scope contract and is never run by oracle/ref_glue_system.cpp.
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include "lvba_unavailable.h"
class SiftGPU {
  public:
    struct SiftKeypoint { float x, y, s, o; };
    enum { SIFTGPU_NOT_SUPPORTED = 0, SIFTGPU_PARTIAL_SUPPORTED = 1, SIFTGPU_FULL_SUPPORTED = 2 };
    SiftGPU(int = 0) {}
    virtual ~SiftGPU() {}
    virtual void ParseParam(int, char **) { lvba_unavailable("SiftGPU::ParseParam"); }
    virtual int CreateContextGL() { lvba_unavailable("SiftGPU::CreateContextGL"); }
    virtual int VerifyContextGL() { lvba_unavailable("SiftGPU::VerifyContextGL"); }
    virtual int RunSIFT(int, int, const void *, unsigned int, unsigned int) { lvba_unavailable("SiftGPU::RunSIFT"); }
    virtual int RunSIFT(const char *) { lvba_unavailable("SiftGPU::RunSIFT"); }
    virtual int GetFeatureNum() { lvba_unavailable("SiftGPU::GetFeatureNum"); }
    virtual void GetFeatureVector(SiftKeypoint *, float *) { lvba_unavailable("SiftGPU::GetFeatureVector"); }
};
class SiftMatchGPU {
  public:
    SiftMatchGPU(int = 4096) {}
    virtual ~SiftMatchGPU() {}
    virtual int VerifyContextGL() { lvba_unavailable("SiftGPU::VerifyContextGL"); }
    virtual void SetMaxSift(int) { lvba_unavailable("SiftGPU::SetMaxSift"); }
    virtual void SetDescriptors(int, int, const float *, int = -1) { lvba_unavailable("SiftGPU::SetDescriptors"); }
    virtual void SetDescriptors(int, int, const unsigned char *, int = -1) { lvba_unavailable("SiftGPU::SetDescriptors"); }
    virtual int GetSiftMatch(int, int (*)[2], float = 0.7f, float = 0.8f, int = 1) { lvba_unavailable("SiftGPU::GetSiftMatch"); }
};
