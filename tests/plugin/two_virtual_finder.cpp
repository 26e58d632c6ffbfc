// A tracker plugin written against the REFERENCE's CameraPoseFinder interface (/root/reference/src/CameraPoseFinder.h:38-39):
// exactly two pure virtuals, initPoseFinder and estimateCameraPose.  It must compile -- and be instantiable -- against
// hybkinectfu_amd/host/hybkf_host.hpp unchanged (tests/test_capi_exports.py builds this file with g++ -fsyntax-only ... and links it).
#include "hybkf_host.hpp"

class CameraPoseFinderConstantVelocity : public CameraPoseFinder {
protected:
  bool initPoseFinder() { _steps = 0; return true; }
  bool estimateCameraPose(const DepthFrameData& depth_frame, const ColorFrameData& color_frame) {
    (void)color_frame;
    if (depth_frame.frameId() == 0) return true;
    ++_steps;
    return true;                          // keeps _pose: a stand-still "tracker"
  }
private:
  int _steps;
};

extern "C" int plugin_two_virtuals_instantiates() {
  CameraPoseFinderConstantVelocity f;     // would not compile if the base class had a third pure virtual
  CameraPoseFinder* base = &f;
  return base->deviceResident() ? 1 : 0;  // a host-side tracker is not device resident
}
