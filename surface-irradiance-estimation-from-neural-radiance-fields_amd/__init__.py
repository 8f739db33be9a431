"""MI355X-native NeRF volume renderer behind the reference's Testbed / pyngp surface.

Sub-modules: scene (layout + camera conventions), synthetic (deterministic scenes), snapshot
(msgpack schema), native (ctypes binding of libngp_hip.so, the C ABI in include/ngp_hip.h),
testbed (host-side mirror of ngp::Testbed), build (hipcc driver).
"""
