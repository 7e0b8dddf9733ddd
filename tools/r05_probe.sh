#!/bin/bash
bash tools/r05_quick_bench.sh
