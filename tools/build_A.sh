#!/bin/bash
# Build the committed HEAD as csrc/build/libA.so (the A arm of tools/ab_bench.sh), then rebuild the working tree.
set -e
cd "$(dirname "$0")/.."
P=adapting-pretrained-vision-transformers-with-lora-against-attack-vectors_amd
git stash -q
$P/csrc/build.sh > /dev/null
cp $P/libvitlora_hip.so $P/csrc/build/libA.so
git stash pop -q
$P/csrc/build.sh
