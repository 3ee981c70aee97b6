"""Copies the reference's two sample scene files (DATA: loader-schema examples,
sampleScenes/teapot_scene.yaml and sampleScenes/shiny_teapot.yaml) into tests/golden/scenes/
byte for byte, because /root/reference does not exist on the GPU box.  Run in the build
container:  python tests/golden/make_scene_fixtures.py
"""
import os
import shutil

REF = "/root/reference/sampleScenes"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for name in ("teapot_scene.yaml", "shiny_teapot.yaml"):
        shutil.copyfile(os.path.join(REF, name), os.path.join(DST, name))
        print("copied", name)
