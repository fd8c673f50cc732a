#!/bin/bash
# Runs ON THE GPU BOX: ExportData.export() into a real file (tools/file_probe.py, 25-snapshot batches of cylinder3D) for the writer's
# thread count and with / without preallocation of a batch's range -- fresh process per setting, two rounds
for round in 1 2; do
  for thr in 1 2 6; do
    for fa in 1 0; do
      S3H5_WRITE_THREADS=$thr S3H5_FALLOCATE=$fa python tools/file_probe.py 25 16 2>&1 | grep "steady" | sed "s/^/fallocate=$fa /" | cut -c1-170
    done
  done
done
