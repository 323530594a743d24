#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python tools/profile_tandem.py 64 2>&1 | grep -E "^==|phase share" | cut -c1-330
