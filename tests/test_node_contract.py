"""Compile-only contract check of the Knowhere node: knowhere_amd/host/hip_index_node.cc is
compiled against the REFERENCE's own headers (include/knowhere/index/index_node.h, config.h, index_factory.h,
index_node_thread_pool_wrapper.h, src/index/ivf/ivf_config.h, src/index/flat/flat_config.h ...), so every override,
Config field, registration macro and Static* signature is checked by the compiler against the interface a Knowhere
maintainer would build it with.  Third-party headers absent from this image (glog, folly, boost iterator_facade, the
milvus-common OpContext / FileManager) come from tests/cpp/ref_stubs/ -- minimal declarations, no reference text.
Skipped where /root/reference is absent (the GPU box); the shim build of the same file is covered by build()."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
JSON_HPP = "/opt/conda/include/json.hpp"

needs_ref = pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "include", "knowhere")) and os.path.exists(JSON_HPP)
                                    and shutil.which("g++")),
                               reason="needs the reference tree, nlohmann json and g++")


def _compile(defs):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror=overloaded-virtual",
           "-DKNHIP_WITH_KNOWHERE_HEADERS", *defs, f"-I{REF}/include", f"-I{REF}/src", f"-I{REF}",
           f"-I{REF}/thirdparty/faiss", f"-I{ROOT}/tests/cpp/ref_stubs", f"-I{ROOT}/include",
           os.path.join(ROOT, "knowhere_amd", "host", "hip_index_node.cc")]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600)


@needs_ref
@pytest.mark.parametrize("defs", [[], ["-DNOT_COMPILE_FOR_SWIG", "-DKNOWHERE_WITH_LIGHT"]],
                         ids=["swig-surface", "light-build"])
def test_node_compiles_against_reference_headers(defs):
    r = _compile(defs)
    assert r.returncode == 0, r.stderr[-4000:]
    assert "error" not in r.stderr


@needs_ref
def test_node_uses_the_reference_interface_not_the_shim():
    """with KNHIP_WITH_KNOWHERE_HEADERS the translation unit must see the reference's IndexNode and never the shim"""
    cmd = ["g++", "-std=c++17", "-E", "-DKNHIP_WITH_KNOWHERE_HEADERS", f"-I{REF}/include", f"-I{REF}/src", f"-I{REF}",
           f"-I{REF}/thirdparty/faiss", f"-I{ROOT}/tests/cpp/ref_stubs", f"-I{ROOT}/include",
           os.path.join(ROOT, "knowhere_amd", "host", "hip_index_node.cc")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "reference/include/knowhere/index/index_node.h" in r.stdout
    assert "reference/include/knowhere/index/index_node_thread_pool_wrapper.h" in r.stdout
    assert "reference/include/knowhere/config.h" in r.stdout
    assert "knowhere_shim.h" not in r.stdout


def test_shim_names_match_reference_macros():
    """the registration the node uses is the reference's macro, by name (index_factory.h)"""
    src = open(os.path.join(ROOT, "knowhere_amd", "host", "hip_index_node.cc")).read()
    assert src.count("KNOWHERE_REGISTER_GLOBAL_WITH_THREAD_POOL(") == 4
    for name in ("GPU_HIP_BRUTE_FORCE", "GPU_HIP_IVF_FLAT", "GPU_HIP_IVF_PQ", "GPU_HIP_IVF_SQ8"):
        assert name in src
    for fn in ("StaticCreateConfig", "StaticHasRawData", "StaticConfigCheck", "checkCancellation",
               "MapSearchResultIdsToOutIds"):
        assert fn in src, fn
    # fp16 / bf16 / int8: the reference's conversion wrapper in front of the fp32 node, one line per index type (compiled
    # by the two tests above: the block is active with the reference's headers)
    assert src.count("KNHIP_MOCK_REGISTER_TYPES(GPU_HIP_") == 4
    assert "IndexNodeDataMockWrapper<data_type>" in src and "MockData<data_type>::type" in src


@needs_ref
def test_typed_registrations_expand_to_the_reference_wrapper():
    """preprocessed with the reference's headers: twelve typed factory entries (4 index types x fp16 / bf16 / int8), each
    building IndexNodeThreadPoolWrapper(IndexNodeDataMockWrapper<T>(node<fp32>))"""
    cmd = ["g++", "-std=c++17", "-E", "-P", "-DKNHIP_WITH_KNOWHERE_HEADERS", f"-I{REF}/include", f"-I{REF}/src", f"-I{REF}",
           f"-I{REF}/thirdparty/faiss", f"-I{ROOT}/tests/cpp/ref_stubs", f"-I{ROOT}/include",
           os.path.join(ROOT, "knowhere_amd", "host", "hip_index_node.cc")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.replace(" ", "").replace("\n", "")
    for t in ("fp16", "bf16", "int8"):
        assert out.count(f"std::make_unique<IndexNodeDataMockWrapper<{t}>>(") == 4, t
        assert out.count(f"MockData<{t}>::type>>(version,object)") == 4, t
