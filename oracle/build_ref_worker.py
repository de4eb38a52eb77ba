"""Recipe: byte-compile the reference's OWN serving files, where they lie, into oracle/_ref/ (test infrastructure, not product).

The reference's model worker (llava/serve/model_worker.py: ModelWorker.__init__ / generate_stream / the FastAPI routes, :45-244) is the caller the
drop-in boundary is written for.  /root/reference does not exist on the GPU box, so the unmodified file could only be imported in
the build container (tests/test_unmodified_worker_boundary.py) and was RE-ENACTED on the GPU (tools/worker_reenactment.py).  This script closes that
gap the way a C reference is handled: it compiles the two files of the path that are the reference's own —

    llava/serve/model_worker.py      the worker
    llava/utils.py                   build_logger / server_error_msg / pretty_print_semaphore, imported by the worker

— with `py_compile`, from /root/reference, into oracle/_ref/llava_pyc/ (sourceless .pyc, importable by the same CPython 3.10 of the GPU image).
oracle/_ref/ is git-ignored (no reference source or derivative enters the history) and not gpurun-ignored (the .pyc files travel to the GPU box like
the built .so).  tests/test_worker_flow_gpu.py::test_unmodified_worker_executes_on_the_gpu imports them there behind the sys.modules aliases of
INTEGRATION.md §A and drives ModelWorker / the /worker_generate_stream route against this build.

    python oracle/build_ref_worker.py [--ref /root/reference]      (also run by __graft_entry__.build() when the reference tree is present)"""
import argparse
import hashlib
import json
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "llava_pyc")
FILES = ("llava/serve/model_worker.py", "llava/utils.py",
         # the model side of the path, for bench.py's cpu_baseline leg of kind "reference" (oracle/ref_shim.py imports these sourceless on the GPU box and times
         # the reference's own forward + greedy loop on the host cores); never imported by the product
         "llava/constants.py", "llava/mm_utils.py", "llava/model/llava_arch.py", "llava/model/language_model/llava_llama.py",
         "llava/model/multimodal_encoder/builder.py", "llava/model/multimodal_encoder/clip_encoder.py", "llava/model/multimodal_projector/builder.py")


def build(ref_root: str = "/root/reference") -> dict:
    manifest = {"python": "%d.%d" % sys.version_info[:2], "files": {}}
    for rel in FILES:
        src = os.path.join(ref_root, rel)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        # dfile: the path tracebacks show — the reference's own path, so a failure on the GPU box points at the file a maintainer knows
        py_compile.compile(src, cfile=dst, dfile=os.path.join("/root/reference", rel), doraise=True)
        with open(src, "rb") as f:
            manifest["files"][rel] = hashlib.sha256(f.read()).hexdigest()
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    return manifest


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.environ.get("LLAVA_REFERENCE_ROOT", "/root/reference"))
    a = ap.parse_args()
    print(json.dumps(build(a.ref), indent=1))
