#!/bin/sh
# Regenerates tests/golden/tf_legacy from the bundles TensorFlow wrote for the reference's legacy model
# (data files, not source): the four index files, and the first 66624 bytes (= the first 8 tensors: encoder layers
# 1 and 2, kernels + biases) of the features data file.  Run in the build container, where /root/reference exists.
set -e
L=/root/reference/.legacy/trained_weights/M4Depth-d6
D=$(dirname "$0")
cp $L/M4Depth/features/checkpoint-200000.index            $D/features.index
cp $L/M4Depth/features/optimizers/checkpoint-200000.index $D/features_optimizers.index
cp $L/M4Depth/upscaler/checkpoint-200000.index            $D/upscaler.index
cp $L/pipeline/checkpoint-200000.index                    $D/pipeline.index
head -c 66624 $L/M4Depth/features/checkpoint-200000.data-00000-of-00001 > $D/features.data-00000-of-00001
chmod u+w $D/*.index $D/features.data-00000-of-00001
