// oracle/embree_tri_probe.cpp -- TEST INFRASTRUCTURE (this container only; links the Embree built by `make -C oracle embree`).
// stdin: nv, nv vertices (x y z), nt, nt index triples, m, m rays (org, dir, tnear, tfar) -- registered as ONE triangle geometry
// exactly as src/pine/impl/accel/embree.cpp:76-87 does.  stdout per ray: primID (-1: miss), tfar, u, v, Ng (hex floats).
#include <embree4/rtcore.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main() {
  int nv = 0, nt = 0, m = 0;
  if (scanf("%d", &nv) != 1) return 1;
  std::vector<float> v(size_t(nv) * 3);
  for (auto& x : v)
    if (scanf("%a", &x) != 1) return 1;
  if (scanf("%d", &nt) != 1) return 1;
  std::vector<unsigned> idx(size_t(nt) * 3);
  for (auto& x : idx)
    if (scanf("%u", &x) != 1) return 1;
  RTCDevice dev = rtcNewDevice(getenv("EMBREE_PROBE_CONFIG"));
  RTCScene scene = rtcNewScene(dev);
  rtcSetSceneFlags(scene, RTC_SCENE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS);
  rtcSetSceneBuildQuality(scene, RTC_BUILD_QUALITY_HIGH);
  RTCGeometry geom = rtcNewGeometry(dev, RTC_GEOMETRY_TYPE_TRIANGLE);
  rtcSetGeometryEnableFilterFunctionFromArguments(geom, true);
  float* vb = (float*)rtcSetNewGeometryBuffer(geom, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, 12, size_t(nv));
  memcpy(vb, v.data(), v.size() * 4);
  unsigned* ib = (unsigned*)rtcSetNewGeometryBuffer(geom, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, 12, size_t(nt));
  memcpy(ib, idx.data(), idx.size() * 4);
  rtcCommitGeometry(geom);
  rtcAttachGeometry(scene, geom);
  rtcReleaseGeometry(geom);
  rtcCommitScene(scene);
  if (scanf("%d", &m) != 1) return 1;
  for (int k = 0; k < m; k++) {
    RTCRayHit rh;
    if (scanf("%a %a %a %a %a %a %a %a", &rh.ray.org_x, &rh.ray.org_y, &rh.ray.org_z, &rh.ray.dir_x, &rh.ray.dir_y, &rh.ray.dir_z, &rh.ray.tnear, &rh.ray.tfar) != 8) return 1;
    rh.ray.mask = unsigned(-1), rh.ray.time = 0.0f, rh.ray.flags = 0;
    rh.hit.geomID = RTC_INVALID_GEOMETRY_ID, rh.hit.primID = RTC_INVALID_GEOMETRY_ID, rh.hit.instID[0] = RTC_INVALID_GEOMETRY_ID;
    RTCIntersectArguments args;
    rtcInitIntersectArguments(&args);
    args.feature_mask = RTCFeatureFlags(RTC_FEATURE_FLAG_TRIANGLE | RTC_FEATURE_FLAG_INSTANCE | RTC_FEATURE_FLAG_FILTER_FUNCTION_IN_ARGUMENTS |
                                        RTC_FEATURE_FLAG_USER_GEOMETRY_CALLBACK_IN_ARGUMENTS);
    rtcIntersect1(scene, &rh, &args);
    if (rh.hit.geomID == RTC_INVALID_GEOMETRY_ID) printf("-1 %a 0x0p+0 0x0p+0 0x0p+0 0x0p+0 0x0p+0\n", rh.ray.tfar);
    else printf("%d %a %a %a %a %a %a\n", int(rh.hit.primID), rh.ray.tfar, rh.hit.u, rh.hit.v, rh.hit.Ng_x, rh.hit.Ng_y, rh.hit.Ng_z);
  }
  rtcReleaseScene(scene);
  rtcReleaseDevice(dev);
  return 0;
}
