"""Extracts the SAM-spec tag table (tag -> sam type, arrow type, description) from the
reference's registry (bio-format-core/src/tag_registry.rs:131-690) into a data fixture.
Run in the build container only (needs /root/reference)."""
import json, re, sys
src = open("/root/reference/datafusion/bio-format-core/src/tag_registry.rs").read()
pat = re.compile(r'tags\.insert\(\s*"(..)"\.to_string\(\),\s*TagDefinition\s*\{\s*sam_type:\s*\'(.)\',\s*arrow_type:\s*([^,]+(?:\([^)]*\))?),\s*description:\s*"((?:[^"\\]|\\.)*)"', re.S)
out = {}
for m in pat.finditer(src):
    tag, st, at, desc = m.groups()
    at = at.strip()
    if at.startswith("DataType::"):
        an = at[len("DataType::"):]
    else:
        mm = re.match(r"list_type\(DataType::(\w+)\)", at)
        an = f"List<{mm.group(1)}>"
    out[tag] = {"sam_type": st, "arrow_type": an, "description": desc.replace('\\"', '"')}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(len(out), "tags")
