#!/bin/bash
TAG=${TAG:-x} bash tools/r06/k10_trace.sh
bash tools/r06/k10_phases.sh | tail -12
