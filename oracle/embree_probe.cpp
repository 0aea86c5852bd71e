// oracle/embree_probe.cpp -- TEST INFRASTRUCTURE (this container only: links the Embree 4.3.1 built from the reference's
// vendored copy by `make -C oracle embree`).  Reads "n" boxes (lower xyz, upper xyz) and "m" rays (org xyz, dir xyz, tnear, tfar,
// then one hit distance per box or a negative number for "no hit") from stdin, registers every box as ONE user primitive exactly
// as src/pine/impl/accel/embree.cpp:88-99 does, and prints for every ray the geometry ids in the order Embree called the
// intersect callback.  tests/ and tools/ compare the restated order (oracle order mode "embree") with it.
#include <embree4/rtcore.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

struct BoxRec {
  float lo[3], hi[3];
};
static std::vector<int> g_log;
static const float* g_hit_t = nullptr;
static void bounds_func(const RTCBoundsFunctionArguments* a) {
  const BoxRec& b = *reinterpret_cast<const BoxRec*>(a->geometryUserPtr);
  a->bounds_o->lower_x = b.lo[0], a->bounds_o->lower_y = b.lo[1], a->bounds_o->lower_z = b.lo[2];
  a->bounds_o->upper_x = b.hi[0], a->bounds_o->upper_y = b.hi[1], a->bounds_o->upper_z = b.hi[2];
}
static void intersect_func(const RTCIntersectFunctionNArguments* a) {
  if (!a->valid[0]) return;
  g_log.push_back(int(a->geomID));
  RTCRayHit* rh = reinterpret_cast<RTCRayHit*>(a->rayhit);
  const float t = g_hit_t[a->geomID];
  if (t >= 0.0f && t > rh->ray.tnear && t < rh->ray.tfar) {
    rh->ray.tfar = t;
    rh->hit.geomID = a->geomID;
    rh->hit.primID = a->primID;
    a->valid[0] = -1;
  }
}
int main() {
  int n = 0, m = 0;
  if (scanf("%d", &n) != 1) return 1;
  std::vector<BoxRec> boxes;
  boxes.resize(size_t(n));
  for (auto& b : boxes)
    if (scanf("%a %a %a %a %a %a", &b.lo[0], &b.lo[1], &b.lo[2], &b.hi[0], &b.hi[1], &b.hi[2]) != 6) return 1;
  RTCDevice dev = rtcNewDevice(getenv("EMBREE_PROBE_CONFIG"));
  RTCScene scene = rtcNewScene(dev);
  rtcSetSceneFlags(scene, RTC_SCENE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS);
  rtcSetSceneBuildQuality(scene, RTC_BUILD_QUALITY_HIGH);
  for (int i = 0; i < n; i++) {
    RTCGeometry g = rtcNewGeometry(dev, RTC_GEOMETRY_TYPE_USER);
    rtcSetGeometryEnableFilterFunctionFromArguments(g, true);
    rtcSetGeometryUserPrimitiveCount(g, 1);
    rtcSetGeometryUserData(g, &boxes[size_t(i)]);
    rtcSetGeometryBoundsFunction(g, bounds_func, nullptr);
    rtcCommitGeometry(g);
    rtcAttachGeometry(scene, g);
    rtcReleaseGeometry(g);
  }
  rtcCommitScene(scene);
  if (scanf("%d", &m) != 1) return 1;
  std::vector<float> hit_t;
  hit_t.resize(size_t(n));
  g_hit_t = hit_t.data();
  for (int k = 0; k < m; k++) {
    RTCRayHit rh;
    if (scanf("%a %a %a %a %a %a %a %a", &rh.ray.org_x, &rh.ray.org_y, &rh.ray.org_z, &rh.ray.dir_x, &rh.ray.dir_y, &rh.ray.dir_z, &rh.ray.tnear, &rh.ray.tfar) != 8) return 1;
    for (auto& t : hit_t)
      if (scanf("%a", &t) != 1) return 1;
    rh.ray.mask = unsigned(-1);
    rh.ray.time = 0.0f;
    rh.ray.flags = 0;
    rh.hit.geomID = RTC_INVALID_GEOMETRY_ID;
    rh.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
    RTCIntersectArguments args;
    rtcInitIntersectArguments(&args);
    args.intersect = intersect_func;
    args.feature_mask = RTCFeatureFlags(RTC_FEATURE_FLAG_TRIANGLE | RTC_FEATURE_FLAG_INSTANCE | RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS |
                                        RTC_FEATURE_FLAG_USER_GEOMETRY_CALLBACK_IN_ARGUMENTS);
    g_log.clear();
    rtcIntersect1(scene, &rh, &args);
    for (int id : g_log) printf("%d ", id);
    printf("| %d %a\n", int(rh.hit.geomID), rh.ray.tfar);
  }
  rtcReleaseScene(scene);
  rtcReleaseDevice(dev);
  return 0;
}
