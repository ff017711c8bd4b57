# the reference's call contracts (per-plane calls), after the printer fix
bench A=1 --per-plane --steps 200
bench A=1 --per-plane --steps 200 --clip 1
bench A=1 --per-plane --steps 200 --clip 1 --frame-sync
bench GFW_COALESCE_PLANES=0 --per-plane --steps 200 --clip 1
