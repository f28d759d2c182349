#!/bin/bash
timeout 200 tools/probes/hop_probe 2>&1 | grep -v "NEVER\|idle"
